#!/usr/bin/env python3
"""Secondary measurements for the non-headline BASELINE configs (not the driver's bench contract):
config 2 (BisectCost, n=1e6, K=32) and config 4 (chunkers with w_max=8 on a banded matrix)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import cpamd
cp = cpamd.load()
from bench import gen_suitesparse_shaped


def banded_dev(n, hb, fill, seed, dev):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    cols, rows = [], []
    for d in range(-hb, hb + 1):
        j = torch.arange(max(0, -d), min(n, n - d), device=dev)
        keep = torch.ones_like(j, dtype=torch.bool) if d == 0 else (torch.rand(j.numel(), generator=g, device=dev) < fill)
        cols.append(j[keep]); rows.append(j[keep] + d)
    cols = torch.cat(cols); rows = torch.cat(rows)
    key = torch.unique(cols * n + rows)
    cols = key // n
    colptr = torch.cat([torch.ones(1, dtype=torch.int64, device=dev), 1 + torch.cumsum(torch.bincount(cols, minlength=n), 0)])
    return colptr.contiguous(), ((key % n) + 1).contiguous()


def timeit(f, reps=3):
    best = 1e30
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n2", type=int, default=1_000_000)
    ap.add_argument("--n4", type=int, default=5_000_000)
    ap.add_argument("--skip-convex", action="store_true", help="the one-wave ConvexTotalChunker kernel takes microseconds per column")
    ap.add_argument("--n4-convex", type=int, default=0, help="size for the ConvexTotalChunker line (default: --n4)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    from chainpartitioners_jl_amd import _lib
    hip = _lib.HipBackend()
    out = {}
    # ---- config 2
    n = args.n2
    colptr, rowval = gen_suitesparse_shaped(n, 13 * n, 0xDEADBEEF + 1, dev)
    h = hip.csr_from_device(n, n, rowval.numel(), colptr.data_ptr(), rowval.data_ptr())
    K = 32
    spl = np.zeros(K + 1, dtype=np.int64)
    for name, mdl in (("work", cp.AffineWorkModel(0, 10, 1)), ("connectivity", cp.AffineConnectivityModel(0, 10, 1, 100))):
        mm = mdl.marshal()
        def run():
            hip.reset_cache(h)
            rc = hip.partition_bisect_cost(h, K, mm, 0.01, 0, spl)
            assert rc == 0, hip.last_error()
        t = timeit(run)
        rc, obj = hip.objective(h, K, spl, mm, None, 1)
        out["cfg2_bisect_" + name] = {"n": n, "nnz": int(rowval.numel()), "K": K, "eps": 0.01, "seconds": t, "bottleneck": obj,
                                      "spl_ok": bool(spl[0] == 1 and spl[-1] == n + 1 and np.all(np.diff(spl) >= 0))}
        if name == "connectivity":                      # SURVEY 8(f) rows on the same input
            spl2 = np.zeros(K + 1, dtype=np.int64)
            def run_lazy():
                hip.reset_cache(h)
                rc = hip.partition_lazy_bisect_cost(h, K, mm, 0.01, spl2)
                assert rc == 0, hip.last_error()
            t = timeit(run_lazy)
            rc, obj2 = hip.objective(h, K, spl2, mm, None, 1)
            out["cfg2_lazy_bisect_connectivity"] = {"seconds": t, "bottleneck": obj2, "same_split_as_bisect_cost": bool(np.array_equal(spl, spl2))}
        spl3 = np.zeros(K + 1, dtype=np.int64)
        def run_index():
            hip.reset_cache(h)
            rc = hip.partition_bisect_index(h, K, mm, 0, spl3)
            assert rc == 0, hip.last_error()
        t = timeit(run_index, reps=1)
        rc, obj3 = hip.objective(h, K, spl3, mm, None, 1)
        out["cfg2_bisect_index_" + name] = {"seconds": t, "bottleneck": obj3}
    hip.csr_destroy(h)
    if args.n4 <= 0:
        print(json.dumps(out))
        return
    # ---- config 4
    from chainpartitioners_jl_amd import api
    fc = cp.ConstrainedCost(cp.ColumnBlockComponentCostModel(3, lambda w: 1 + w), cp.VertexCount(), 8)
    for name, fname, n in (("convex", "pack_convex", args.n4_convex or args.n4), ("dynamic", "pack_dynamic", args.n4)):
        if name == "convex" and args.skip_convex:
            continue
        colptr, rowval = banded_dev(n, 16, 0.5, 0xDEADBEEF + 4, dev)
        h = hip.csr_from_device(n, n, rowval.numel(), colptr.data_ptr(), rowval.data_ptr())
        class _Shape:            # the host mirror sizes closure tables from the matrix shape only
            pass
        sh = _Shape(); sh.n = n; sh.m = n
        _, mm, wm, wi, wf, _, keep = api._marshal(sh, fc, None, stack_method=(name == "convex"))
        fn = getattr(hip, fname)
        splc = np.zeros(n + 1, dtype=np.int64); Kout = np.zeros(1, dtype=np.int64)
        def run():
            hip.reset_cache(h)
            rc = fn(h, mm, None, wm, wi, wf, splc, Kout)
            assert rc == 0, hip.last_error()
        t = timeit(run, reps=1)
        Kc = int(Kout[0])
        out["cfg4_pack_" + name] = {"n": n, "nnz": int(rowval.numel()), "w_max": 8, "seconds": t, "us_per_column": t / n * 1e6, "chunks": Kc,
                                    "width_ok": bool(np.all(np.diff(splc[:Kc + 1]) <= 8) and splc[Kc] == n + 1)}
        hip.csr_destroy(h)
        del colptr, rowval
    print(json.dumps(out))


if __name__ == "__main__":
    main()
