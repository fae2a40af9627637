"""Row-tiled DP (the multi-GPU path, chainpartitioners.jl_amd/distributed.py) on ONE GPU: G ranks are simulated in
one process -- every rank computes only its tile of each layer into the shared layer buffer, exactly the data flow the
RCCL all_gather implements -- and the split vector must equal the single-rank result and the CPU oracle."""
import numpy as np
import pytest
import torch

from util import cp, sprand, golden_matrices, suitesparse_shaped

pytestmark = pytest.mark.gpu


def run_simulated(hip, A, K, method, world):
    from chainpartitioners_jl_amd.distributed import TiledDP
    dev = torch.device("cuda", 0)
    h = hip.csr(A)
    ranks = [TiledDP(hip, h, A.n, K, method, g, world, dev) for g in range(world)]
    try:
        shared_prev = ranks[0].prev
        shared_cur = ranks[0].cur
        for T in ranks:                                   # all ranks share the two layer buffers (= the gathered vectors)
            T.prev, T.cur = shared_prev, shared_cur
        ranks[0].step_layer(1)
        for T in ranks:
            T.swap()
        for k in range(2, K + 1):
            for T in ranks:
                T.step_layer(k)                           # writes only its own tile of `cur`
            for T in ranks:
                T.swap()
        spl = np.zeros(K + 1, dtype=np.int64)
        spl[K] = A.n + 1
        for k in range(K, 0, -1):
            spl[k - 1] = max(T.ptr_at(k, int(spl[k])) for T in ranks)       # the MAX all_reduce
        return spl
    finally:
        for T in ranks:
            T.close()


def test_tiled_equals_single_rank_and_oracle(hip, orc):
    rng = np.random.default_rng(77)
    mats = [sprand(8, 16, 0.3, rng), sprand(20, 40, 0.1, rng), sprand(9, 65, 0.2, rng), golden_matrices()["HB/can_292"],
            golden_matrices()["LPnetlib/lp_etamacro"], suitesparse_shaped(3000, 6, 5)]
    for A in mats:
        for K in (2, 3, 5):
            for mdl in (cp.AffineConnectivityModel(0, 10, 1, 100), cp.AffineConnectivityModel(0, 0, 0, 1),
                        cp.AffineHyperedgeCutModel(0, 0, 0, 0, 1), cp.AffineConnectivityModel(0, 3, 1, 3, alpha_k=list(range(1, K + 1)))):
                meth = cp.DynamicTotalSplitter(mdl)
                want = cp.partition_stripe(A, K, meth, backend=orc).spl
                for world in (1, 2, 3, 8):
                    got = run_simulated(hip, A, K, meth, world)
                    assert np.array_equal(got, want), (A, K, mdl.kind, world)
            # bottleneck objective: general sweep, tiled the same way
            meth = cp.DynamicBottleneckSplitter(cp.AffineConnectivityModel(0, 3, 1, 3))
            if A.n <= 1000:
                want = cp.partition_stripe(A, K, meth, backend=orc).spl
                assert np.array_equal(run_simulated(hip, A, K, meth, 3), want)


def _tiled_worker(rank, world, port, q):
    import os, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cpamd
    cpm = cpamd.load()
    from chainpartitioners_jl_amd import _lib
    from chainpartitioners_jl_amd.distributed import partition_stripe_tiled
    from util import suitesparse_shaped as ss
    hipb = _lib.HipBackend(0)
    A = ss(4000, 6, 11)
    K = 6
    meth = cpm.DynamicTotalSplitter(cpm.AffineConnectivityModel(0, 10, 1, 100))
    spl = partition_stripe_tiled(hipb, hipb.csr(A), A.n, K, meth, device=torch.device("cuda", 0))
    if rank == 0:
        q.put(spl.tolist())
    dist.barrier()
    dist.destroy_process_group()


def test_tiled_two_processes_gloo(hip, orc):
    """Two real processes (both on this box's single GPU; collectives rehearsed over gloo, as RCCL needs one GPU
    per rank): same driver code path as bench.py --mode tiled."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tiled_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    A = suitesparse_shaped(4000, 6, 11)
    want = cp.partition_stripe(A, 6, cp.DynamicTotalSplitter(cp.AffineConnectivityModel(0, 10, 1, 100)), backend=orc).spl
    assert got == want.tolist()
