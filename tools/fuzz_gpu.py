#!/usr/bin/env python3
"""Differential fuzz on the GPU box: random shapes / models / K, HIP path vs CPU oracle, bit-exact split vectors.
Usage: python tools/fuzz_gpu.py [seconds] [seed]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
from util import cp, sprand, suitesparse_shaped, banded
from chainpartitioners_jl_amd import _lib
import orc_binding

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
hip = _lib.HipBackend(); orc = orc_binding.OracleBackend()


def rand_matrix():
    kind = rng.integers(0, 5)
    if kind == 0:
        m, n = int(rng.integers(1, 40)), int(rng.integers(1, 60))
        return sprand(m, n, float(rng.uniform(0.0, 0.6)), rng)
    if kind == 1:
        return suitesparse_shaped(int(rng.integers(50, 2500)), int(rng.integers(2, 12)), int(rng.integers(1 << 30)))
    if kind == 2:
        return banded(int(rng.integers(50, 2500)), int(rng.integers(1, 20)), float(rng.uniform(0.1, 0.9)), int(rng.integers(1 << 30)))
    if kind == 3:                                    # a few very heavy columns and runs of empty ones
        n = int(rng.integers(100, 3000)); m = int(rng.integers(100, 3000))
        deg = rng.integers(0, 4, n); deg[rng.integers(0, n, 3)] = m // 2; deg[n // 3: n // 3 + n // 10] = 0
        colptr = np.concatenate([[1], 1 + np.cumsum(deg)]).astype(np.int64)
        rows = np.concatenate([np.sort(rng.choice(m, size=int(d), replace=False)) + 1 for d in deg if d > 0] or [np.zeros(0)]).astype(np.int64)
        return cp.SparseMatrixCSC(m, n, colptr, rows)
    m, n = int(rng.integers(200, 800)), int(rng.integers(1000, 3000))     # wide: many columns per row
    return sprand(m, n, 0.01, rng)


def rand_model():
    k = rng.integers(0, 9)
    r = lambda lo, hi: int(rng.integers(lo, hi))
    if k == 0: return cp.AffineConnectivityModel(r(-3, 4), r(-3, 4), r(-2, 3), r(0, 5))
    if k == 1: return cp.AffineConnectivityModel(0, 0, 0, 1)
    if k == 2: return cp.AffineWorkModel(r(-3, 4), r(-3, 11), r(-2, 3))
    if k == 3: return cp.AffineHyperedgeCutModel(r(-2, 3), r(-2, 3), r(-2, 3), r(-2, 2), r(2, 5))
    if k == 4: return cp.AffineConnectivityModel(float(r(-3, 4)), 0.0, 1.0, float(r(0, 4)))
    if k == 5: return cp.AffineHyperedgeCutModel(0, 0, 0, 0, 1)
    if k == 6: return cp.AffineConnectivityModel(0.5, 0.25, 1.0, 3.0)    # non-integral: general sweep (total), valley search (bottleneck)
    if k == 7: return cp.AffineHyperedgeCutModel(r(0, 3), r(0, 3), r(0, 3), r(2, 6), r(0, 3))            # b_self >= b_cut >= 0: bottleneck valley class
    return cp.AffineHyperedgeCutModel(0.0, 0.5, 0.0, 0.7, 0.1)           # non-integral hyperedge: OUTSIDE the valley class (general sweep)


t0 = time.time(); cases = 0
while time.time() - t0 < budget:
    A = rand_matrix(); f = rand_model(); K = int(rng.choice([1, 2, 3, 5, 8, 17]))
    tag = (A.m, A.n, A.nnz, type(f).__name__, f._params(), K)
    meths = [cp.DynamicTotalSplitter(f), cp.DynamicTotalChunker(f)]
    hyp_valley = isinstance(f, cp.AffineHyperedgeCutModel) and f.dtype == cp.models.CP_I64 and f._params()[3] >= f._params()[4] >= 0 and all(p >= 0 for p in f._params()[1:3])
    if A.n <= 1500 or hyp_valley or (all(p >= 0 for p in f._params()[1:]) and not isinstance(f, cp.AffineHyperedgeCutModel)):
        meths += [cp.DynamicBottleneckSplitter(f), cp.DynamicBottleneckChunker(f)]          # (valley search for growing costs at any n -- the wave walks; general sweep below 1500)
    if isinstance(f, (cp.AffineConnectivityModel, cp.AffineWorkModel)) and all(p >= 0 for p in f._params()) and sum(f._params()) > 0:
        meths += [cp.BisectCostBottleneckSplitter(f, 0.01), cp.BisectIndexBottleneckSplitter(f)]
        if isinstance(f, cp.AffineConnectivityModel):
            meths += [cp.LazyBisectCostBottleneckSplitter(f, 0.05)]
    for meth in meths:
        got = cp.partition_stripe(A, K, meth, backend=hip)
        want = cp.partition_stripe(A, K, meth, backend=orc)
        assert got == want, ("partition", type(meth).__name__, tag)
    # the width-constrained K-part DP (windowed path for the inverse-Monge models, one-wave kernel otherwise), both loop orders;
    # windows from "barely feasible" to "wider than the matrix"
    if A.n >= 1 and A.n <= 2500:
        for w in (max(1, -(-A.n // K) + int(rng.integers(0, 4))), max(1, int(rng.integers(1, A.n + 3)))):
            # the width weight under its own name, or as a work model alpha + c * width (windowed path too: csrc/capi.hip width_of_weight)
            wk = int(rng.integers(0, 4))
            c = int(rng.integers(1, 4)); a0 = int(rng.integers(0, 3))
            # wk 3: a pin-weighted budget (the bottleneck DP takes it on the valley search through the weight's j0 array)
            fc = (cp.ConstrainedCost(f, cp.VertexCount(), w) if wk == 0 else
                  cp.ConstrainedCost(f, cp.AffineWorkModel(a0, c, 0), a0 + c * w + int(rng.integers(0, c))) if wk == 1 else
                  cp.ConstrainedCost(f, cp.AffineWorkModel(0.25 * a0, 0.5 * c, 0.0), 0.25 * a0 + 0.5 * c * w + 0.1) if wk == 2 else
                  cp.ConstrainedCost(f, cp.AffineWorkModel(a0, c - 1, 1), a0 + (c - 1) * w + int(w * max(A.nnz, 1) / max(A.n, 1)) + int(rng.integers(0, 5))))
            for meth in (cp.DynamicTotalSplitter(fc), cp.DynamicTotalChunker(fc), cp.DynamicBottleneckSplitter(fc), cp.DynamicBottleneckChunker(fc)):
                if wk == 3 and A.n > 600 and meth.combine == 0:
                    continue                       # (total cost under a pin weight: the one-wave literal kernel, 40 s at n = 6000)
                if meth.order == 1 and getattr(f, "alpha_k", None) is not None:
                    continue
                got = cp.partition_stripe(A, K, meth, backend=hip)
                want = cp.partition_stripe(A, K, meth, backend=orc)
                assert got == want, ("constrained", type(meth).__name__, w, tag)
    if A.n >= 1:
        for w in (int(rng.integers(1, 20)),):
            fc = cp.ConstrainedCost(f, cp.VertexCount(), w)
            for meth in (cp.DynamicTotalChunker(fc), cp.ConvexTotalChunker(fc), cp.ConcaveTotalChunker(fc)):
                if A.n > 3000 and not isinstance(meth, cp.DynamicTotalChunker):
                    continue
                got = cp.pack_stripe(A, meth, backend=hip)
                want = cp.pack_stripe(A, meth, backend=orc)
                assert got == want, ("pack", type(meth).__name__, w, tag)
    # the batch entry points against their own loops
    if cases % 7 == 0 and A.n >= 2:
        wm = cp.AffineWorkModel(0, int(rng.integers(1, 5)), 1); nm = cp.AffineConnectivityModel(0, int(rng.integers(0, 5)), 1, int(rng.integers(1, 9)))
        reqs = [(int(rng.choice([1, 2, 5, 9])), cp.BisectCostBottleneckSplitter(m, float(rng.choice([0.1, 0.01])))) for m in (wm, nm, wm, nm)]
        for (Kb, m), g in zip(reqs, cp.partition_stripe_batch(A, reqs, backend=hip)):
            assert g == cp.partition_stripe(A, Kb, m, backend=orc), ("bisect batch", tag)
        cm = [cp.ConvexTotalChunker(cp.ConstrainedCost(cp.ColumnBlockComponentCostModel(int(rng.integers(1, 5)), lambda x: 1 + x), cp.VertexCount(), int(rng.integers(1, 16)))) for _ in range(3)]
        if A.n <= 3000:
            for m, g in zip(cm, cp.pack_stripe_batch(A, cm, backend=hip)):
                assert g == cp.pack_stripe(A, m, backend=orc), ("convex batch", tag)
    cases += 1
    if cases % 5 == 0:
        print("  %d cases, %.0f s" % (cases, time.time() - t0), flush=True)
print("fuzz ok:", cases, "cases in %.0f s, seed %d" % (time.time() - t0, seed))
