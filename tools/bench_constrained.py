#!/usr/bin/env python3
"""Timing of the width-constrained DP at bench size (the reference's own script: bin/test_table_constrained_splits.jl:28,
DynamicTotalSplitter(ConstrainedCost(AffineConnectivityModel(0,0,0,1), VertexCount(), ceil(1.5 n / K)))) with the per-kernel
breakdown.  bench.py --config constrained prints the driver-visible line; this is the developer's view."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import cpamd
cp = cpamd.load()
from bench import gen_suitesparse_shaped


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--deg", type=int, default=10)
    ap.add_argument("--parts", type=int, default=64)
    ap.add_argument("--wfac", type=float, default=1.5)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--opt", action="append", default=[])
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    from chainpartitioners_jl_amd import _lib
    hip = _lib.HipBackend()
    n, K = args.n, args.parts
    colptr, rowval = gen_suitesparse_shaped(n, args.deg * n, 0xDEADBEEF + 2, dev)
    N = int(rowval.numel())
    h = hip.csr_from_device(n, n, N, colptr.data_ptr(), rowval.data_ptr())
    mdl = cp.AffineConnectivityModel(0, 0, 0, 1)
    mm = mdl.marshal(); wm = cp.VertexCount().marshal()
    w = int(np.ceil(args.wfac * n / K))
    spl = np.zeros(K + 1, dtype=np.int64)
    for kv in args.opt:
        k, v = kv.split("="); assert hip.set_option(k, int(v)) == 0

    def step():
        hip.reset_cache(h)
        rc = hip.partition_dynamic(h, K, 0, 0, mm, None, wm, w, float(w), spl)
        assert rc == 0, (rc, hip.last_error())
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    hip.prof_reset(); hip.prof_enable(True)
    step()
    torch.cuda.synchronize()
    hip.prof_enable(False)
    prof = {k: round(v["ms"], 2) for k, v in hip.prof_get().items() if v["launches"]}
    rc, obj = hip.objective(h, K, spl, mm, None, 0)
    widths = np.diff(spl)
    print(json.dumps({"n": n, "nnz": N, "K": K, "w_max": w, "seconds": dt, "objective": int(obj), "max_width": int(widths.max()),
                      "nonempty_parts": int((widths > 0).sum()), "spl_head": spl[:6].tolist(), "kernels_ms": prof}))


if __name__ == "__main__":
    main()
