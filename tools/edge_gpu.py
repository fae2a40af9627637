#!/usr/bin/env python3
"""Edge shapes on the GPU box: zero columns / zero rows / K > n, every method, HIP vs oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
from util import cp
from chainpartitioners_jl_amd import _lib
import orc_binding
hip = _lib.HipBackend(); orc = orc_binding.OracleBackend()
def both(fn, tag):
    out = []
    for b in (hip, orc):
        try:
            r = fn(b); out.append(("ok", r.spl.tolist() if hasattr(r, "spl") else r))
        except Exception as e:
            out.append((type(e).__name__, ""))
    print(tag, out[0], "==" if out[0] == out[1] else "!=", out[1], flush=True)
    return out[0] == out[1]
ok = True
shapes = [(0, 0), (3, 0), (0, 3), (1, 1), (2, 5)]
for (m, n) in shapes:
    colptr = np.ones(n + 1, dtype=np.int64); rows = np.zeros(0, dtype=np.int64)
    A = cp.SparseMatrixCSC(m, n, colptr, rows)
    for K in (1, 2, 7):
        for f in (cp.AffineConnectivityModel(0, 3, 1, 3), cp.AffineWorkModel(1, 10, 1), cp.AffineHyperedgeCutModel(0, 1, 1, 1, 3)):
            for meth in (cp.DynamicTotalSplitter(f), cp.DynamicBottleneckSplitter(f), cp.DynamicTotalChunker(f), cp.ConvexTotalSplitter(f), cp.ConcaveTotalSplitter(f)):
                ok &= both(lambda b: cp.partition_stripe(A, K, meth, backend=b), (m, n, K, type(f).__name__, type(meth).__name__))
            if not isinstance(f, cp.AffineHyperedgeCutModel):
                for meth in (cp.BisectCostBottleneckSplitter(f, 0.01), cp.BisectIndexBottleneckSplitter(f)) + ((cp.LazyBisectCostBottleneckSplitter(f, 0.01),) if isinstance(f, cp.AffineConnectivityModel) else ()):
                    ok &= both(lambda b: cp.partition_stripe(A, K, meth, backend=b), (m, n, K, type(f).__name__, type(meth).__name__))
    for f in (cp.AffineConnectivityModel(0, 3, 1, 3), cp.AffineWorkModel(1, 10, 1)):
        for fc in (f, cp.ConstrainedCost(f, cp.VertexCount(), 2)):
            for meth in (cp.DynamicTotalChunker(fc), cp.ConvexTotalChunker(fc), cp.ConcaveTotalChunker(fc)):
                ok &= both(lambda b: cp.pack_stripe(A, meth, backend=b), (m, n, "pack", type(meth).__name__, isinstance(fc, cp.ConstrainedCost)))
print("ALL EQUAL" if ok else "MISMATCHES")
