#!/usr/bin/env python3
"""bench.py -- benchmarks of the partition_stripe / pack_stripe hot path on MI355X.

  python bench.py [--config C] --gpus N --steps K --warmup W

N > 1: either under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (RANK / WORLD_SIZE in the
environment) or plain `python bench.py --gpus N`, which starts the N ranks itself from a parent that never touches a GPU
(launch_ranks, before `import torch`).  WORLD_SIZE != --gpus is an error.  The line carries `value` = independent partitions, one
per GPU (weak scaling, no data-path collective) and `extras.tiled` = ONE partition with its DP rows tiled over the GPUs and an
RCCL all_gather per layer (strong scaling); `--mode tiled` makes the row-tiled partition `value` itself.

Configs (every matrix comes from the seeded SplitMix64 generator tests/synth.py, base seed 0xDEADBEEF +
config index; one "step" = one full partition call INCLUDING oracle construction, colptr / rowval already resident in HBM --
what the reference's `@benchmarkable partition_stripe($A,$K,$f)` times, test/runbenchmarks.jl:65):

  3 (default, the metric of BASELINE.json)  DynamicTotalSplitter(AffineConnectivityModel{Int64}(0,0,0,1)), K = 64, on
              suitesparse_shaped n = m = 10^7, nnz = 10^8.  Its answer is closed-form ([1, n+1, ..., n+1]: for every cost the
              O(n log^2 n) path admits the diagonal candidate ties the minimum); all K layers are computed.
  constrained DynamicTotalSplitter(ConstrainedCost(the same cost, VertexCount(), ceil(1.5 n / K))) -- the reference's own script
              (bin/test_table_constrained_splits.jl:28) -- on the same matrix: a non-trivial answer on the windowed path.
  constrained-bottleneck  DynamicBottleneckSplitter(ConstrainedCost(AffineConnectivityModel(0,10,1,100), w, w_max)) on the same matrix:
              --weight width: VertexCount(), w_max = ceil(1.5 n / K) (the factor of bin/test_table_constrained_splits.jl:28);
              --weight pins: AffineWorkModel(0,0,1), w_max = ceil(1.5 nnz / K).  Valley search with candidate limits (csrc/dp_bottleneck.hip).
  bottleneck  DynamicBottleneckSplitter(AffineConnectivityModel(0,10,1,100)), K = 64, same matrix; cross-checked at full size
              against BisectIndexBottleneckSplitter (exact).
  2           BisectCostBottleneckSplitter(AffineWorkModel(0,10,1) / AffineConnectivityModel(0,10,1,100), 0.01), n = 10^6, K = 32.
  4           ConvexTotalChunker / DynamicTotalChunker(ConstrainedCost(ColumnBlockComponentCostModel{Int}(3, w->1+w),
              VertexCount(), 8)) on banded n = 5*10^6 (test/runbenchmarks.jl:21,31-33), the oracle timed beside it at full size.
  5           DynamicTotalSplitter(AffineHyperedgeCutModel(0,0,0,0,1)), K = 256, on n = 5*10^7, nnz = 5*10^8 (BASELINE config 5;
              `--gpus 8 --mode tiled` is its 8-GPU row-tiled form, `--gpus 1` the same partition on one GPU: ptr is 4 B x K x n = 51 GB).
  5shape      the same with K = 64 (round-2 comparisons).

Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline     : the dominant kernel, HIP-event timed inside the timed region, against the 8 TB/s HBM peak
  cpu_baseline : the literal CPU restatement (oracle/, kind "port", 1 core) on a bounded sample
and, for the default config, `extras.other_configs`: short runs of `constrained`, `bottleneck` and `2`, the time with the
host-to-device copy of the pattern, and the second CPU baseline (the oracle's ConvexTotalSplitter).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


# ------------------------------------------------------------------------------------------------ N > 1: the launcher
def launch_ranks(gpus, argv):
    """`python bench.py --gpus N` with no RANK in the environment: start N worker processes, one per device, from THIS process,
    which has touched no GPU (it runs before `import torch`; the workers are children, nothing is re-exec'ed).  Rank 0's stdout
    is ours, so its ONE JSON line is the launcher's line.  Any worker that fails takes the group down and the launcher fails."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(gpus), LOCAL_WORLD_SIZE=str(gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            c = p.poll()
            if c is None:
                continue
            alive.remove(p)
            if c != 0 and rc == 0:
                rc = c
                sys.stderr.write("bench.py launcher: rank %d exited with %d; stopping the other ranks\n" % (procs.index(p), c))
                for q in alive:
                    q.terminate()                     # (the exact children started above)
    return rc


def _early_gpus(argv):
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return 1


if __name__ == "__main__" and "RANK" not in os.environ and _early_gpus(sys.argv[1:]) > 1:
    sys.exit(launch_ranks(_early_gpus(sys.argv[1:]), sys.argv[1:]))

import numpy as np
import torch

import cpamd
import synth

HBM_PEAK_GBS = 8000.0     # MI355X spec HBM3E peak, MI355X_MICROARCH.md "Chip-level parameters"
SEED = 0xDEADBEEF


def gen_suitesparse_shaped(n, N, seed, device):
    """`suitesparse_shaped` (tests/synth.py) trimmed to exactly N nonzeros: 1-based int64 colptr / rowval on `device`"""
    _, _, colptr, rowval = synth.suitesparse_shaped_t(n, N / n, seed, device, nnz=N)
    return colptr, rowval


def oracle_backend():
    sys.path[:0] = [os.path.join(ROOT, "oracle")]
    import orc_binding
    return orc_binding.OracleBackend()


def measure_copy_gbs(dev):
    """Device-to-device copy bandwidth of this box (bytes read + bytes written per second), the practical HBM ceiling."""
    a = torch.empty(1 << 28, dtype=torch.int32, device=dev)      # 1 GiB
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    del a, b
    return 2.0 * (1 << 30) / (ms * 1e-3) / 1e9


def timed(f, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


# ------------------------------------------------------------------------------------------------ CPU baselines (oracle, 1 core)
def cpu_fit_quadratic(make_method, K, mean_deg, seed, sizes=(2000, 4000, 8000), repeats=3, budget_s=60.0, per_k=True):
    """the literal O(n^2)-per-layer sweeps cannot run at bench size: time them at three sizes on the same generator (min of
    `repeats` runs each; the sizes are the largest three that fit the budget: n = 4*10^4 alone would take four minutes),
    fit t = a * [K *] n^2 by least squares through the origin and extrapolate.  -> (a, [(n, seconds, a_n)])"""
    cp = cpamd.load()
    orc = oracle_backend()
    samples, used = [], 0.0
    for n in sizes:
        est = samples[-1][1] * 4 * repeats if samples else 0.0
        if samples and used + est > budget_s:
            break
        _, _, colptr, rowval = synth.suitesparse_shaped_np(n, mean_deg, seed, nnz=int(n * mean_deg))
        A = cp.SparseMatrixCSC(n, n, colptr, rowval)
        best = None
        for _ in range(repeats):
            t0 = time.perf_counter()
            cp.partition_stripe(A, K, make_method(cp, n), backend=orc)
            dt = time.perf_counter() - t0
            used += dt
            best = dt if best is None else min(best, dt)
        samples.append((n, best, best / ((K if per_k else 1) * float(n) * n)))
    x = np.array([(K if per_k else 1) * float(n) * n for n, _, _ in samples])
    y = np.array([t for _, t, _ in samples])
    return float((x * y).sum() / (x * x).sum()), samples


def cpu_convex_splitter(K, mean_deg, seed, hip, n=100_000):
    """SURVEY 8(d) second baseline: the oracle's ConvexTotalSplitter -- the reference's own fastest exact-value method for this
    cost -- timed directly at the largest n that fits the budget, and the GPU's DynamicTotalSplitter on the SAME matrix beside it
    (pattern resident, oracle structures rebuilt, like `value`) so that the ratio is taken at one size.
    -> (n, oracle seconds, gpu seconds, same total value)"""
    cp = cpamd.load()
    orc = oracle_backend()
    _, _, colptr, rowval = synth.suitesparse_shaped_np(n, mean_deg, seed, nnz=int(n * mean_deg))
    A = cp.SparseMatrixCSC(n, n, colptr, rowval)
    mdl = cp.AffineConnectivityModel(0, 0, 0, 1)
    t0 = time.perf_counter()
    P = cp.partition_stripe(A, K, cp.ConvexTotalSplitter(mdl), backend=orc)
    tc = time.perf_counter() - t0
    G = cp.partition_stripe(A, K, cp.DynamicTotalSplitter(mdl), backend=hip)      # (uploads the pattern, warms the handle)

    def gpu():
        hip.reset_cache(A)
        cp.partition_stripe(A, K, cp.DynamicTotalSplitter(mdl), backend=hip)
    tg = min(timed(gpu, 1) for _ in range(3))
    same = cp.total_value(A, P, mdl, backend=hip) == cp.total_value(A, G, mdl, backend=hip)
    return n, tc, tg, bool(same)


# ------------------------------------------------------------------------------------------------ the DP configs on one matrix
class DpBench:
    """config 3 / constrained / bottleneck / 5shape: K-part DP on a suitesparse_shaped pattern resident in HBM"""

    def __init__(self, args, cfg, dev, rank, world, tiled):
        self.cp = cpamd.load()
        from chainpartitioners_jl_amd import _lib
        self.hip = _lib.HipBackend(device=dev.index)
        self.args, self.cfg, self.dev, self.tiled = args, cfg, dev, tiled
        big = cfg in ("5", "5shape")
        self.n = args.n or (50_000_000 if big else 10_000_000)
        N = args.nnz or (10 * self.n if args.n else (500_000_000 if big else 100_000_000))
        self.K = args.parts or (256 if cfg == "5" else 64)
        seed = SEED + (5 if big else 3) - 1 + (0 if tiled else 1000 * rank)      # config index; independent partitions: one matrix per rank
        self.colptr, self.rowval = gen_suitesparse_shaped(self.n, N, seed, dev)
        self.N = int(self.rowval.numel())
        torch.cuda.synchronize()
        self.h = self.hip.csr_from_device(self.n, self.n, self.N, self.colptr.data_ptr(), self.rowval.data_ptr())
        cp = self.cp
        self.combine, self.wm, self.w, self.wmodel, self.wname = 0, None, 0, None, ""
        if cfg in ("3", "constrained"):
            self.mdl = cp.AffineConnectivityModel(0, 0, 0, 1); self.mname = "AffineConnectivityModel{Int64}(0,0,0,1)"
        elif cfg in ("bottleneck", "constrained-bottleneck"):
            self.mdl = cp.AffineConnectivityModel(0, 10, 1, 100); self.mname = "AffineConnectivityModel{Int64}(0,10,1,100)"; self.combine = 1
        else:
            self.mdl = cp.AffineHyperedgeCutModel(0, 0, 0, 0, 1); self.mname = "AffineHyperedgeCutModel{Int64}(0,0,0,0,1)"
        if cfg == "constrained":
            self.wmodel, self.wname = cp.VertexCount(), "VertexCount()"; self.w = -(-3 * self.n // (2 * self.K))
        elif cfg == "constrained-bottleneck":
            if getattr(args, "weight", "width") == "pins":
                self.wmodel, self.wname = cp.AffineWorkModel(0, 0, 1), "AffineWorkModel{Int64}(0,0,1)"; self.w = -(-3 * self.N // (2 * self.K))
            else:
                self.wmodel, self.wname = cp.VertexCount(), "VertexCount()"; self.w = -(-3 * self.n // (2 * self.K))
        if self.wmodel is not None:
            self.wm = self.wmodel.marshal()
        self.mm = self.mdl.marshal()
        self.spl = np.zeros(self.K + 1, dtype=np.int64)

    def method_name(self):
        if self.wmodel is not None:
            return "Dynamic%sSplitter(ConstrainedCost(%s, %s, %d))" % ("Bottleneck" if self.combine else "Total", self.mname, self.wname, self.w)
        return ("DynamicBottleneckSplitter(%s)" if self.combine else "DynamicTotalSplitter(%s)") % self.mname

    def step(self):
        hip, h = self.hip, self.h
        hip.reset_cache(h)           # every step rebuilds the oracle structures, as one reference call does
        if self.tiled:
            from chainpartitioners_jl_amd.distributed import partition_stripe_tiled
            cost = self.cp.ConstrainedCost(self.mdl, self.wmodel, self.w) if self.wmodel is not None else self.mdl
            meth = self.cp.DynamicBottleneckSplitter(cost) if self.combine else self.cp.DynamicTotalSplitter(cost)
            self.spl[:] = partition_stripe_tiled(hip, h, self.n, self.K, meth, device=self.dev)
            return
        rc = hip.partition_dynamic(h, self.K, self.combine, 0, self.mm, None, self.wm, self.w, float(self.w), self.spl)
        if rc != 0:
            raise RuntimeError(f"cp_partition_dynamic -> {rc}: {hip.last_error()}")

    def check(self):
        """size-independent properties at full size (bit-exact parity itself is established by tests/ at oracle sizes)"""
        n, K, spl, hip = self.n, self.K, self.spl, self.hip
        assert spl[0] == 1 and spl[-1] == n + 1 and np.all(np.diff(spl) >= 0), spl
        rc, obj = hip.objective(self.h, K, spl, self.mm, None, self.combine)
        assert rc == 0
        rc, whole = hip.objective(self.h, 1, np.array([1, n + 1], dtype=np.int64), self.mm, None, self.combine)
        info = {"objective": int(obj)}
        if self.cfg == "constrained":
            assert int(np.diff(spl).max()) <= self.w                         # every part respects w_max
            equi = np.minimum(1 + self.w * np.arange(K + 1, dtype=np.int64), n + 1)      # a feasible partition: greedy full-width parts
            rc, obj_eq = hip.objective(self.h, K, equi, self.mm, None, 0)
            assert obj <= obj_eq                                              # the optimum is no worse than a feasible partition
            info.update({"max_width": int(np.diff(spl).max()), "w_max": int(self.w), "nonempty_parts": int((np.diff(spl) > 0).sum()),
                         "objective_of_full_width_parts": int(obj_eq)})
        elif self.combine:
            bi = np.zeros(K + 1, dtype=np.int64)
            rc = hip.partition_bisect_index(self.h, K, self.mm, 0, bi)
            assert rc == 0
            rc, obj_bi = hip.objective(self.h, K, bi, self.mm, None, 1)
            info["bisect_index_bottleneck"] = int(obj_bi)
            if self.wmodel is None:
                assert obj == obj_bi, (obj, obj_bi)                           # the DP optimum == the exact BisectIndex optimum
            else:
                # every part within its budget; no better than the unconstrained optimum; no worse than a feasible partition
                # (greedy parts filled to the budget, the columns' pin counts read back from the resident colptr)
                colptr = self.colptr.cpu().numpy().astype(np.int64)
                wgt = (np.diff(spl) if self.wname == "VertexCount()" else colptr[spl[1:] - 1] - colptr[spl[:-1] - 1])
                assert int(wgt.max()) <= self.w, (int(wgt.max()), self.w)
                assert obj >= obj_bi
                if self.wname == "VertexCount()":
                    greedy = np.minimum(1 + self.w * np.arange(K + 1, dtype=np.int64), n + 1)
                else:
                    greedy = np.ones(K + 1, dtype=np.int64)
                    for k in range(1, K + 1):                     # the furthest column whose pins still fit
                        greedy[k] = min(n + 1, int(np.searchsorted(colptr, colptr[greedy[k - 1] - 1] + self.w, side="right")))
                assert greedy[-1] == n + 1
                rc, obj_gr = hip.objective(self.h, K, greedy, self.mm, None, 1)
                assert obj <= obj_gr, (obj, obj_gr)
                info.update({"max_part_weight": int(wgt.max()), "w_max": int(self.w), "objective_of_greedy_full_parts": int(obj_gr),
                             "unconstrained_answer_feasible": bool(obj == obj_bi)})
        else:
            assert obj >= whole                                               # sum_k nets_k >= nets(all columns)
        return info


def time_region(B, steps, dist, dev):
    """EXACTLY `steps` steps bracketed by barrier + synchronize on both sides; the MAX over the ranks"""
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        B.step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    return dt


def tiled_extra(args, cfg, dev, rank, world, dist, out, expect_s):
    """N > 1, default mode: after the independent partitions (`value`, weak scaling), ONE partition of the same shape with its
    DP rows tiled over the ranks (strong scaling, chainpartitioners.jl_amd/distributed.py) goes into `extras.tiled`.  The
    row-tiled exchange has met one RCCL rank only before the driver's scaling run, so a watchdog turns a stuck or failed run
    into an `error` entry: the weak-scaling line is printed regardless."""
    steps = max(1, min(args.steps, 3))
    limit = 180.0 + 8.0 * (steps + 1) * expect_s
    res = {"n_gpus": world, "scaling": "strong", "steps": steps, "warmup": 1, "mode": "one partition, DP rows tiled over the GPUs, one all_gather per layer"}
    done = threading.Event()

    def bail(msg):
        if rank == 0:
            res["error"] = msg
            out["extras"]["tiled"] = res
            print(json.dumps(out), flush=True)
        os._exit(0)

    def watchdog():
        if not done.wait(limit):
            bail("no answer from the row-tiled run within %.0f s" % limit)
    threading.Thread(target=watchdog, daemon=True).start()
    try:
        B = DpBench(args, cfg, dev, rank, world, True)
        B.step()
        dt = time_region(B, steps, dist, dev)
        res.update({"ms_per_step": dt / steps * 1e3, "value": steps / dt, "unit": "partitions/s", "check": B.check()})
        B.hip.csr_destroy(B.h)
    except Exception as e:                               # (the other ranks are waiting in a collective: their watchdogs end them)
        bail("row-tiled run failed on rank %d: %r" % (rank, e))
    done.set()
    if rank == 0:
        out["extras"]["tiled"] = res


def dp_cpu_baseline(B, cfg, sizes):
    """the oracle's literal DP of the same method on one host core, fitted at small n and extrapolated"""
    n, N, K = B.n, B.N, B.K
    Kfit = min(K, 64)                    # (the law is linear in K: a is fitted with at most 64 parts, stated in `sample`)
    avg_deg = N / n
    if cfg == "constrained-bottleneck":
        pins = B.wname != "VertexCount()"
        mk = lambda cp, nn: cp.DynamicBottleneckSplitter(cp.ConstrainedCost(B.mdl, B.wmodel, -(-3 * (int(nn * avg_deg) if pins else nn) // (2 * Kfit))))
        a, samples = cpu_fit_quadratic(mk, Kfit, avg_deg, SEED + 2, sizes=sizes, per_k=False)
        t_full = a * float(n) * float(n)
        law = ("t = a*n^2 with a=%.3e s (as for the constrained total DP: the rows of layer k's window, each over at most one budget of candidates "
               "-- Theta(n w K / 2) = Theta(n^2) steps whatever K)" % a)
    elif cfg == "constrained":
        mk = lambda cp, nn: cp.DynamicTotalSplitter(cp.ConstrainedCost(B.mdl, cp.VertexCount(), -(-3 * nn // (2 * Kfit))))
        a, samples = cpu_fit_quadratic(mk, Kfit, avg_deg, SEED + 2, sizes=sizes, per_k=False)
        t_full = a * float(n) * float(n)
        law = ("t = a*n^2 with a=%.3e s (layer k scans its window of up to (n+1) - (K-k) w .. k w rows, each over at most w = ceil(1.5 n / K) "
               "candidates: Theta(n w K / 2) = Theta(n^2) steps whatever K)" % a)
    else:
        mk = (lambda cp, nn: cp.DynamicBottleneckSplitter(B.mdl)) if B.combine else (lambda cp, nn: cp.DynamicTotalSplitter(B.mdl))
        a, samples = cpu_fit_quadratic(mk, Kfit, avg_deg, SEED + 2, sizes=sizes)
        t_full = a * K * float(n) * float(n)
        law = "t = a*K*n^2 with a=%.3e s" % a
    return {"value": 1.0 / t_full, "unit": "partitions/s", "cores": 1, "kind": "port",
            "sample": "literal restatement of %s timed at n=%s (K=%d, same generator, min of 3 runs each), least-squares %s, extrapolated to n=%d, K=%d"
                      % (B.method_name(), [s[0] for s in samples], Kfit, law, n, K),
            "sample_seconds": [s[1] for s in samples], "a_per_size": [s[2] for s in samples], "host_cores": os.cpu_count()}


def run_dp(args, cfg, dev, rank, world, dist):
    tiled = args.mode == "tiled" and dist is not None
    B = DpBench(args, cfg, dev, rank, world, tiled)
    hip, n, N, K = B.hip, B.n, B.N, B.K
    for kv in args.opt:
        name, val = kv.split("=")
        assert hip.set_option(name, int(val)) == 0, kv
    copy_gbs = measure_copy_gbs(dev)
    # The per-kernel breakdown (HIP events around every launch group: ~40 records per DP round) is taken on an UNTIMED step; the
    # timed region keeps events around the dominant kernel only (2 per round), as the roofline line needs them live.
    for _ in range(max(args.warmup - 1, 0)):
        B.step()
    hip.prof_reset(); hip.prof_enable(True)
    B.step()
    torch.cuda.synchronize()
    prof_all = hip.prof_get()
    slots = list(prof_all.keys())
    cands = (("dp_lpass_own", "k_lpass_own"), ("dp_lpass", "k_lpass"), ("dp_brute", "k_bn_walk"))
    dom, kname = max(cands, key=lambda kv: prof_all.get(kv[0], {"ms": 0.0})["ms"])
    hip.prof_reset()
    assert hip.set_option("prof_only", slots.index(dom)) == 0
    dt = time_region(B, args.steps, dist, dev)
    B.last_step_s = dt / args.steps
    hip.prof_enable(False)
    hip.set_option("prof_only", -1)
    prof = hip.prof_get()
    spl_t = torch.from_numpy(B.spl.copy()).to(dev)
    if dist is not None:
        gathered = [torch.empty_like(spl_t) for _ in range(world)]      # the K+1-entry split vectors of all ranks: one RCCL all_gather
        dist.all_gather(gathered, spl_t)
    info = B.check()
    if rank != 0:
        return None, B
    ms_per_step = dt / args.steps * 1e3
    value = (1 if tiled else world) * args.steps / dt
    ex = prof[dom]
    bytes_per_launch = ex["units"] / max(ex["launches"], 1)       # accumulated by the library per launch
    avg_ms = ex["ms"] / max(ex["launches"], 1)
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    b_alg = 8.0 * (n + 1 + N) + K * (n + 1) * 24.0 + 8.0 * (K + 1)      # SURVEY.md 8(d) whole-partition bytes
    # HBM bytes per launch from the PMC counters: a STORED measurement (separate --pmc passes cannot run inside this process,
    # tools/pmc_lpass.sh), so it is printed with its source and the commit it was taken at -- null when none exists for this kernel
    traffic = None
    for rnd in ("r03", "r02"):
        pmc_path = os.path.join(ROOT, "profiles", "%s_pmc_%s.json" % (rnd, kname))
        if cfg == "3" and os.path.exists(pmc_path) and (n, N) == (10_000_000, 100_000_000):
            pm = json.load(open(pmc_path))
            traffic = {"bytes": pm.get("traffic_bytes_per_launch_corrected"), "bytes_raw": pm.get("traffic_bytes_per_launch_raw"),
                       "stored": True, "source": "profiles/%s_pmc_%s.json" % (rnd, kname), "measured_at": pm.get("measured_at", "round 2 (commit 5d697b7)")}
            break
    out = {
        "metric": "partitions/sec, %s, K=%d" % (B.method_name(), K),
        "value": value, "unit": "partitions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if tiled else "weak", "vs_baseline": None,
        "dtype": "int64", "data": "synthetic",
        "config": {"workload": "config %s: %s on suitesparse_shaped CSR (seeded SplitMix64 generator, tests/synth.py), n=%d rows, nnz=%d, K=%d; %s"
                               % (cfg, B.method_name(), n, N, K, "one partition, DP rows tiled over the GPUs" if tiled else "one independent partition per GPU"),
                   "n": n, "nnz": N, "K": K, "includes_oracle_build": True,
                   "device_block_pool": True},      # (every step rebuilds the oracle structures; the raw device blocks they live in are recycled: csrc/core.hip dev_alloc)
        "roofline": {"bound": "hbm", "kernel": "%s (%s)" % (dom, kname), "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "measured_copy_gbs": copy_gbs,      # this box's device-to-device copy rate (read + write bytes), SURVEY 8(d)
                     "avg_launch_ms": avg_ms, "launches_per_step": ex["launches"] / args.steps,
                     "alg_bytes_per_launch": bytes_per_launch,
                     "whole_path": {"alg_bytes": b_alg, "achieved": b_alg / (ms_per_step * 1e-3) / 1e9,
                                    "frac": b_alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS}},
        "kernels_ms_per_step": {k: round(v["ms"], 3) for k, v in prof_all.items() if v["launches"]},      # from the untimed profiling step
        "check": info,
        "extras": {},
    }
    if cfg == "3":
        out["extras"]["closed_form_note"] = ("for every cost this path admits, ptr[j',k] = j' and spl = [1, n+1, ..., n+1] (DESIGN.md section 4): "
                                             "`value` times all K layers; the layer kernels are checked with injected previous rows (tests/test_gpu_blocks.py)")
        if not tiled:
            # untimed extras, never part of `value`
            assert hip.set_option("fixed_point", 1) == 0
            keep = B.spl.copy()
            out["extras"]["ms_per_step_with_fixed_point_exit"] = timed(B.step, 1) * 1e3
            hip.set_option("fixed_point", 0)
            assert np.array_equal(B.spl, keep), "fixed-point exit changed the partition"
            # the same call with the pattern in host memory: cp_csr_create copies colptr / rowval over PCIe first
            hc, hr = B.colptr.cpu().numpy(), B.rowval.cpu().numpy()
            A_host = B.cp.SparseMatrixCSC(n, n, hc, hr)

            def with_h2d():
                meth = B.cp.DynamicTotalSplitter(B.mdl)
                B.cp.partition_stripe(A_host, K, meth, backend=hip)
            out["extras"]["ms_per_step_with_h2d"] = timed(with_h2d, 1) * 1e3
            del A_host, hc, hr
    if not args.no_cpu_baseline and world == 1:          # (rank 0 at N = 1 only)
        out["cpu_baseline"] = dp_cpu_baseline(B, cfg, sizes=getattr(args, "cpu_sizes", (2000, 4000, 8000)))
    return out, B


# ------------------------------------------------------------------------------------------------ config 2
def run_cfg2(args, dev, brief=False):
    cp = cpamd.load()
    from chainpartitioners_jl_amd import _lib
    hip = _lib.HipBackend(device=dev.index)
    n = args.n or 1_000_000
    K = args.parts or 32
    _, _, colptr, rowval = synth.suitesparse_shaped_t(n, 13, SEED + 1, dev, nnz=13 * n)
    N = int(rowval.numel())
    h = hip.csr_from_device(n, n, N, colptr.data_ptr(), rowval.data_ptr())
    spl = np.zeros(K + 1, dtype=np.int64)
    res = {}
    for name, mdl in (("work", cp.AffineWorkModel(0, 10, 1)), ("connectivity", cp.AffineConnectivityModel(0, 10, 1, 100))):
        mm = mdl.marshal()

        def run():
            hip.reset_cache(h)
            rc = hip.partition_bisect_cost(h, K, mm, 0.01, 0, spl)
            assert rc == 0, hip.last_error()
        run()
        t = timed(run, max(args.steps, 3))
        rc, obj = hip.objective(h, K, spl, mm, None, 1)
        rc, lo, hi = hip.bound_stripe(h, K, mm)
        assert lo <= obj <= hi                                               # the bounds sandwich of test_Costs.jl
        res[name] = {"ms_per_step": t * 1e3, "bottleneck": int(obj), "bound_stripe": [int(lo), int(hi)]}
    # ---- a batch: 256 requests (K x eps x model constants) on the same pattern in ONE launch (cp_partition_bisect_cost_batch): the probe
    # chain of one partition fills one wave; a sweep fills the chip and shares the counting structure
    breq = []
    for Kb in (8, 16, 32, 64, 128, 256, 512, 1024):
        for eps in (0.1, 0.03, 0.01, 0.003):
            for mdl in (cp.AffineWorkModel(0, 10, 1), cp.AffineWorkModel(0, 1, 1), cp.AffineConnectivityModel(0, 10, 1, 100), cp.AffineConnectivityModel(0, 0, 0, 1),
                        cp.AffineConnectivityModel(0, 1, 1, 10), cp.AffineConnectivityModel(5, 10, 1, 100), cp.AffineConnectivityModel(0, 3, 1, 30), cp.AffineConnectivityModel(0, 100, 1, 1000)):
                breq.append((Kb, mdl, eps))
    bmms = [m.marshal() for _, m, _ in breq]

    def run_batch():
        hip.reset_cache(h)
        rc, spls = hip.partition_bisect_cost_batch(h, [k for k, _, _ in breq], bmms, [e for _, _, e in breq], [0] * len(breq))
        assert rc == 0, hip.last_error()
        return spls
    bspl = run_batch()
    tb = timed(run_batch, max(args.steps, 3))
    # the same requests one by one (what a loop over cp_partition_bisect_cost costs, structures rebuilt per call as a reference call does)
    def run_loop():
        for (kb, _, e), mm in list(zip(breq, bmms))[:32]:
            hip.reset_cache(h)
            sp = np.zeros(kb + 1, dtype=np.int64)
            assert hip.partition_bisect_cost(h, kb, mm, e, 0, sp) == 0
    run_loop()
    tl = timed(run_loop, 1) / 32
    for i in (0, 37, 101, 255):                              # spot check against the single call
        kb, _, e = breq[i]
        sp = np.zeros(kb + 1, dtype=np.int64)
        assert hip.partition_bisect_cost(h, kb, bmms[i], e, 0, sp) == 0 and np.array_equal(sp, bspl[i]), i
    batch = {"batch": len(breq), "ms_total": tb * 1e3, "ms_per_problem": tb * 1e3 / len(breq), "ms_per_problem_one_by_one": tl * 1e3,
             "requests": "K in {8..1024} x eps in {0.1, 0.03, 0.01, 0.003} x 8 Work / Connectivity models; link arrays + ONE net counter included"}
    hip.csr_destroy(h)
    info = {"n": n, "nnz": N, "K": K, "eps": 0.01, "results": res, "batch": batch}
    if brief:
        return info
    # CPU baseline: the oracle's BisectCost directly at full size
    out = {"metric": "partitions/sec, BisectCostBottleneckSplitter(AffineConnectivityModel{Int64}(0,10,1,100), 0.01), K=%d" % K,
           "value": 1e3 / res["connectivity"]["ms_per_step"], "unit": "partitions/s", "n_gpus": 1, "steps": max(args.steps, 3), "warmup": 1,
           "ms_per_step": res["connectivity"]["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
           "data": "synthetic",
           "config": {"workload": "config 2: BisectCostBottleneckSplitter on suitesparse_shaped CSR, n=%d, nnz=%d, K=%d, eps=0.01 (includes link-array + counter build)" % (n, N, K)},
           "roofline": {"bound": "hbm", "kernel": "k_bisect (one wave: latency-bound probe chain; roofline N/A, SURVEY 8d)", "achieved": 8.0 * (n + 1 + N) / (res["connectivity"]["ms_per_step"] * 1e-3) / 1e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 8.0 * (n + 1 + N) / (res["connectivity"]["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None},
           "check": info}
    if not args.no_cpu_baseline:
        orc = oracle_backend()
        A = cp.SparseMatrixCSC(n, n, colptr.cpu().numpy(), rowval.cpu().numpy())
        t0 = time.perf_counter()
        P = cp.partition_stripe(A, K, cp.BisectCostBottleneckSplitter(cp.AffineConnectivityModel(0, 10, 1, 100), 0.01), backend=orc)
        tc = time.perf_counter() - t0
        assert np.array_equal(P.spl, spl), "GPU and oracle split vectors differ at full size"
        out["cpu_baseline"] = {"value": 1.0 / tc, "unit": "partitions/s", "cores": 1, "kind": "port",
                               "sample": "the oracle's BisectCost at the full size n=%d (direct; same split vector as the GPU)" % n, "host_cores": os.cpu_count()}
        # the batch against the oracle: four of its requests, timed one by one (each builds its own structures, as a reference call does)
        tcb = 0.0
        for i in (3, 66, 130, 250):
            kb, mdl, e = breq[i]
            t0 = time.perf_counter()
            P = cp.partition_stripe(A, kb, cp.BisectCostBottleneckSplitter(mdl, e), backend=orc)
            tcb += time.perf_counter() - t0
            assert np.array_equal(P.spl, bspl[i]), "batch and oracle split vectors differ (request %d)" % i
        batch["oracle_ms_per_problem"] = tcb / 4 * 1e3
        batch["speedup_per_problem_vs_oracle_1_core"] = (tcb / 4) / (tb / len(breq))
    return out


# ------------------------------------------------------------------------------------------------ config 4
def run_cfg4(args, dev):
    cp = cpamd.load()
    from chainpartitioners_jl_amd import _lib
    hip = _lib.HipBackend(device=dev.index)
    n = args.n or 5_000_000
    _, _, colptr, rowval = synth.banded_t(n, 16, 0.5, SEED + 4, dev)
    N = int(rowval.numel())
    h = hip.csr_from_device(n, n, N, colptr.data_ptr(), rowval.data_ptr())
    colb = cp.ColumnBlockComponentCostModel(3, lambda w: 1 + w)
    f = cp.ConstrainedCost(colb, cp.VertexCount(), 8)
    import types
    from chainpartitioners_jl_amd import api
    proxy = types.SimpleNamespace(n=n, m=n)             # (the closure w -> 1 + w is tabulated for the widths a method evaluates)
    mms = {"convex": api._marshal(proxy, f, None, stack_method=True), "dynamic": api._marshal(proxy, f, None)}
    out_spl = {}
    times = {}
    for name in ("convex", "dynamic"):
        spl = np.zeros(n + 2, dtype=np.int64); Kout = np.zeros(1, dtype=np.int64)
        _, mm, wm, wi, wf, _, _keep = mms[name]

        def run():
            hip.reset_cache(h)
            rc = (hip.pack_convex if name == "convex" else hip.pack_dynamic)(h, mm, None, wm, wi, wf, spl, Kout)
            assert rc == 0, hip.last_error()
        run()
        times[name] = timed(run, max(1, args.steps if name == "dynamic" else 1))
        out_spl[name] = spl[:int(Kout[0]) + 1].copy()
        assert int(np.diff(out_spl[name]).max()) <= 8
    value = {}
    mm_obj = api._marshal(proxy, colb, None)[1]
    for name in ("convex", "dynamic"):
        s = out_spl[name]
        rc, v = hip.objective(h, len(s) - 1, s, mm_obj, None, 0)
        assert rc == 0, hip.last_error()
        value[name] = int(v)
    assert value["dynamic"] <= value["convex"]          # the DP is optimal; the convex chunker is a heuristic on this non-Monge cost
    out = {"metric": "partitions/sec, pack_stripe(ConvexTotalChunker(ConstrainedCost(ColumnBlockComponentCostModel{Int}(3, w->1+w), VertexCount(), 8)))",
           "value": 1.0 / times["convex"], "unit": "partitions/s", "n_gpus": 1, "steps": 1, "warmup": 1, "ms_per_step": times["convex"] * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
           "config": {"workload": "config 4: ConvexTotalChunker + ColumnBlock cost, w_max=8, banded n=%d (half bandwidth 16, fill 0.5), nnz=%d; includes link arrays + window table" % (n, N)},
           "roofline": {"bound": "hbm", "kernel": "k_pack_convex_win (one wave, sequential by the algorithm: latency-bound)",
                        "achieved": (8.0 * (n + 1 + N) + 24.0 * (n + 1)) / times["convex"] / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": (8.0 * (n + 1 + N) + 24.0 * (n + 1)) / times["convex"] / 1e9 / HBM_PEAK_GBS, "traffic": None},
           "check": {"K_convex": len(out_spl["convex"]) - 1, "K_dynamic": len(out_spl["dynamic"]) - 1, "total_value": value,
                     "dynamic_total_chunker_ms": times["dynamic"] * 1e3}}
    # ---- a batch: requests (cost constants x width limits) on the same pattern in ONE launch (cp_pack_convex_batch), one wave each
    nb = args.batch
    bmeth = []
    for i in range(nb):
        a = 1 + i % 16
        w = 4 + (i // 16) % 5                                # widths 4 .. 8
        bmeth.append((cp.ColumnBlockComponentCostModel(a, lambda x: 1 + x), w))
    bmms = [api._marshal(proxy, cp.ConstrainedCost(m, cp.VertexCount(), w), None, stack_method=True)[1] for m, w in bmeth]
    hip.reset_cache(h)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc, bspl = hip.pack_convex_batch(h, bmms, [w for _, w in bmeth], n)
    tb = time.perf_counter() - t0
    assert rc == 0, hip.last_error()
    for (m, w), sp in zip(bmeth, bspl):
        assert sp[0] == 1 and sp[-1] == n + 1 and int(np.diff(sp).max()) <= w
    i0 = 4 * 16 + 2                                          # a = 1 + i % 16 = 3, w = 4 + i // 16 = 8: the request of the single call above
    if i0 < nb:
        assert np.array_equal(bspl[i0], out_spl["convex"]), "the batch's request (3, w -> 1 + w, 8) differs from the single call"
    out["check"]["batch"] = {"batch": nb, "s_total": tb, "ms_per_problem": tb * 1e3 / nb,
                             "requests": "ColumnBlockComponentCostModel(a, w -> 1 + w), a in 1..16, VertexCount limits 4..8; link arrays, ONE net counter, ONE table of net counts, "
                                         "the copies of the chunk vectors to the host and their unravelling included"}
    if not args.no_cpu_baseline:
        orc = oracle_backend()
        A = cp.SparseMatrixCSC(n, n, colptr.cpu().numpy(), rowval.cpu().numpy())
        t0 = time.perf_counter()
        P = cp.pack_stripe(A, cp.ConvexTotalChunker(f), backend=orc)
        tc = time.perf_counter() - t0
        assert np.array_equal(P.spl, out_spl["convex"]), "GPU and oracle chunk vectors differ at full size"
        out["cpu_baseline"] = {"value": 1.0 / tc, "unit": "partitions/s", "cores": 1, "kind": "port", "seconds": tc,
                               "sample": "the oracle's ConvexTotalChunker at the full size n=%d (direct; identical chunk vector)" % n, "host_cores": os.cpu_count()}
        # one more request of the batch against the oracle
        ib = min(nb - 1, 7 * 16 + 5)
        mb, wb = bmeth[ib]
        t0 = time.perf_counter()
        Pb = cp.pack_stripe(A, cp.ConvexTotalChunker(cp.ConstrainedCost(mb, cp.VertexCount(), wb)), backend=orc)
        tcb = time.perf_counter() - t0
        assert np.array_equal(Pb.spl, bspl[ib]), "batch and oracle chunk vectors differ"
        out["check"]["batch"]["oracle_s_per_problem"] = (tc + tcb) / 2
        out["check"]["batch"]["speedup_per_problem_vs_oracle_1_core"] = ((tc + tcb) / 2) / (tb / nb)
    hip.csr_destroy(h)
    return out


def dry_run(args, rank, world, dist):
    """--dry-run (launcher / protocol rehearsal on a box without GPUs, tests/test_bench_launcher.py): the rendezvous, the
    barrier-bracketed timed region, the MAX over ranks and the split-vector all_gather of the real run over gloo, with a
    stand-in step that computes nothing.  `value` is null: this is not a measurement."""
    def step():
        time.sleep(0.01 * (1 + rank))
    for _ in range(args.warmup):
        step()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    dist.barrier()
    tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    me = torch.tensor([rank, os.getpid()], dtype=torch.int64)
    seen = [torch.empty_like(me) for _ in range(world)]
    dist.all_gather(seen, me)
    if rank != 0:
        return None
    return {"metric": "dry run of the N-rank bench protocol (no GPU work)", "value": None, "unit": "partitions/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(tmax.item()) / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "none", "dry_run": True,
            "config": {"workload": "dry run"}, "ranks_seen": [int(t[0]) for t in seen], "pids": [int(t[1]) for t in seen]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="3", choices=["3", "constrained", "bottleneck", "constrained-bottleneck", "2", "4", "5", "5shape"])
    ap.add_argument("--weight", choices=["width", "pins"], default="width", help="config constrained-bottleneck: the weight of the ConstrainedCost")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=0)
    ap.add_argument("--nnz", type=int, default=0)
    ap.add_argument("--parts", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="config 3: skip the short runs of the other configs; N > 1: skip extras.tiled")
    ap.add_argument("--opt", action="append", default=[], help="library tunable name=value (cp_set_option), e.g. short_t=4")
    ap.add_argument("--mode", choices=["independent", "tiled"], default="independent",
                    help="N>1: 'independent' = one partition per GPU (weak scaling, `value`) followed by ONE row-tiled partition in "
                         "extras.tiled (strong scaling); 'tiled' = `value` itself is the row-tiled partition (RCCL all_gather per layer)")
    ap.add_argument("--batch", type=int, default=256, help="config 4: requests of the batched ConvexTotalChunker line")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="gloo: only with --dry-run")
    ap.add_argument("--dry-run", action="store_true", help="rehearse launcher + collectives without GPU work (tests)")
    ap.add_argument("--emit-spl", action="store_true", help="put rank 0's split vector into check.spl (parity tests at small n)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE is %d: launch with `python bench.py --gpus N` (self-launching) or "
                 "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`" % (args.gpus, world))
    if args.backend == "gloo" and not args.dry_run:
        sys.exit("bench.py: the gloo backend exists for --dry-run only; measurements run over RCCL")
    dist = None
    if args.dry_run:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        out = dry_run(args, rank, world, dist)
        if out is not None:
            print(json.dumps(out), flush=True)
        dist.destroy_process_group()
        return
    if world > 1 or args.mode == "tiled":
        # (--mode tiled on one GPU: a one-rank RCCL group, so that the row-tiled driver and its collectives can be timed on a one-GPU box)
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        # RCCL 2.26 prints a version banner on rank 0's stdout when its communicator comes up (a C printf, flushed at once): stdout
        # carries the ONE JSON line, so file descriptor 1 points at stderr until the group exists and has run its first collective
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    if args.config == "2":
        out = run_cfg2(args, dev) if rank == 0 else None
    elif args.config == "4":
        out = run_cfg4(args, dev) if rank == 0 else None
    else:
        out, B = run_dp(args, args.config, dev, rank, world, dist)
        if out is not None and args.emit_spl:
            out["check"]["spl"] = [int(x) for x in B.spl]
        if world > 1 and args.mode == "independent" and not args.no_extras:
            expect_s = float(B.last_step_s)
            B.hip.csr_destroy(B.h); del B
            torch.cuda.empty_cache()
            tiled_extra(args, args.config, dev, rank, world, dist, out, expect_s)
        elif out is not None and args.config == "3" and world == 1 and not args.no_extras:
            # driver-visible lines of the other single-GPU configs: short runs on the same resident matrix
            other = {}
            sub = argparse.Namespace(**vars(args)); sub.steps = 5; sub.warmup = 1; sub.cpu_sizes = (2000, 4000)      # (sub-second configs: a few more steps, the clocks of a short run vary)
            B.hip.csr_destroy(B.h); del B
            torch.cuda.empty_cache()
            for c in ("constrained", "bottleneck"):
                o, b2 = run_dp(sub, c, dev, 0, 1, None)
                other[c] = {"metric": o["metric"], "ms_per_step": o["ms_per_step"], "value": o["value"], "check": o["check"],
                            "dominant_kernel": o["roofline"]["kernel"], "roofline_frac": o["roofline"]["frac"], "kernels_ms_per_step": o["kernels_ms_per_step"]}
                if "cpu_baseline" in o:
                    other[c]["cpu_baseline"] = o["cpu_baseline"]
                b2.hip.csr_destroy(b2.h); del b2
                torch.cuda.empty_cache()
            sub.n = 0; sub.parts = 0
            other["2"] = run_cfg2(sub, dev, brief=True)
            out["extras"]["other_configs"] = other
            if not args.no_cpu_baseline:
                from chainpartitioners_jl_amd import _lib
                nn, tc, tg, same = cpu_convex_splitter(out["config"]["K"], out["config"]["nnz"] / out["config"]["n"], SEED + 2, _lib.HipBackend(device=dev.index))
                out["extras"]["cpu_baseline_2"] = {"method": "ConvexTotalSplitter(AffineConnectivityModel(0,0,0,1)) (the reference's fastest exact-value method for this cost), oracle, 1 core",
                                                   "n": nn, "K": out["config"]["K"], "seconds": tc,
                                                   "gpu_seconds_same_matrix": tg, "gpu_method": "DynamicTotalSplitter, all K layers, structures rebuilt",
                                                   "same_total_value": same,
                                                   "note": "timed directly at n=%d (same generator); super-linear in n (wavelet queries leave cache): n=1e7 is out of reach" % nn}
    if rank == 0 and out is not None:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
