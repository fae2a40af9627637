"""GPU parity for the SURVEY 8(f) rows: [Flip]BisectIndexBottleneckSplitter and LazyBisectCostBottleneckSplitter
(connectivity specialisation) through the C ABI vs the CPU oracle: bit-exact split vectors."""
import numpy as np
import pytest

from util import cp, sprand, golden_matrices, suitesparse_shaped, banded

pytestmark = pytest.mark.gpu


def _mats(seed):
    rng = np.random.default_rng(seed)
    mats = [sprand(m, n, 0.3, rng) for m in (1, 3, 8) for n in (1, 2, 3, 8, 40)] + list(golden_matrices().values())
    return mats


def test_bisect_index_matches_oracle(hip, orc):
    rng = np.random.default_rng(3)
    for A in _mats(31) + [suitesparse_shaped(20000, 8, 6)]:
        for K in (1, 2, 3, 8, 32):
            if A.n > 5000 and K > 8:
                continue
            for f in (cp.AffineWorkModel(0, 10, 1), cp.AffineConnectivityModel(0, 10, 1, 100), cp.AffineConnectivityModel(0, 3, 1, 3),
                      cp.AffineConnectivityModel(0.5, 0.25, 1.0, 3.0), cp.AffineWorkModel(2.5, 0.5, 1.25),
                      cp.AffineConnectivityModel(0, 3, 1, 3, alpha_k=rng.integers(1, 11, K).tolist())):
                for meth in (cp.BisectIndexBottleneckSplitter(f), cp.FlipBisectIndexBottleneckSplitter(f)):
                    got = cp.partition_stripe(A, K, meth, backend=hip)
                    want = cp.partition_stripe(A, K, meth, backend=orc)
                    assert got == want, (A, K, f.kind, f.dtype, meth.flip)
            # decreasing costs for the Flip variant (test_Partitioners.jl:116-152)
            base = 1 + A.nnz + 3 * A.n + 3 * A.m
            f = cp.AffineConnectivityModel(0, -3, -1, -3, alpha_k=(base + rng.integers(1, 11, K)).tolist())
            got = cp.partition_stripe(A, K, cp.FlipBisectIndexBottleneckSplitter(f), backend=hip)
            want = cp.partition_stripe(A, K, cp.FlipBisectIndexBottleneckSplitter(f), backend=orc)
            assert got == want, (A, K, "flip funky")


def test_bisect_index_is_optimal(hip):
    """test_Partitioners.jl:101,112 with eps = 0: same bottleneck as the exact DP (computed on the device)."""
    A = suitesparse_shaped(3000, 6, 2)
    for f in (cp.AffineWorkModel(0, 10, 1), cp.AffineConnectivityModel(0, 10, 1, 100)):
        for K in (2, 7):
            opt = cp.partition_stripe(A, K, cp.DynamicBottleneckSplitter(f), backend=hip)
            got = cp.partition_stripe(A, K, cp.BisectIndexBottleneckSplitter(f), backend=hip)
            assert cp.bottleneck_value(A, got, f, backend=hip) == cp.bottleneck_value(A, opt, f, backend=hip)


def test_lazy_bisect_cost_matches_oracle(hip, orc):
    rng = np.random.default_rng(5)
    big = [suitesparse_shaped(20000, 8, 6), suitesparse_shaped(60000, 3, 7), banded(30000, 16, 0.5, 8)]
    # one very heavy column (longer than a 16 Ki-entry chunk) and runs of empty columns
    n = 3000
    deg = np.zeros(n, dtype=np.int64); deg[5] = 40000; deg[100:200] = 7; deg[2500:] = 3
    colptr = np.concatenate([[1], 1 + np.cumsum(deg)]).astype(np.int64)
    rows = np.concatenate([np.sort(rng.choice(50000, size=d, replace=False)) + 1 for d in deg if d > 0]).astype(np.int64)
    big.append(cp.SparseMatrixCSC(50000, n, colptr, rows))
    for A in _mats(32) + big:
        for K in (1, 2, 3, 8, 32):
            for f in (cp.AffineConnectivityModel(0, 10, 1, 100), cp.AffineConnectivityModel(0, 3, 1, 3), cp.AffineConnectivityModel(0, 0, 0, 1),
                      cp.AffineConnectivityModel(0.5, 0.25, 1.0, 3.0),
                      cp.AffineConnectivityModel(0, 3, 1, 3, alpha_k=rng.integers(1, 11, K).tolist())):
                for eps in (0.1, 0.01, 0.001):
                    meth = cp.LazyBisectCostBottleneckSplitter(f, eps)
                    got = cp.partition_stripe(A, K, meth, backend=hip)
                    want = cp.partition_stripe(A, K, meth, backend=orc)
                    assert got == want, (A, K, f.dtype, eps)


def test_lazy_rejects_work_models(hip):
    A = suitesparse_shaped(100, 4, 1)
    with pytest.raises(AssertionError):
        cp.partition_stripe(A, 2, cp.LazyBisectCostBottleneckSplitter(cp.AffineWorkModel(0, 10, 1), 0.1), backend=hip)


def test_concave_methods_match_oracle(hip, orc):
    """ConcaveTotalChunker / ConcaveTotalSplitter: identical control flow on the device (bit-exact split vectors), for
    concave costs and -- the algorithm is deterministic for any cost -- for the affine and connectivity models too."""
    w = cp.AffineWorkModel(0, 1, 0)
    mats = _mats(33) + [suitesparse_shaped(3000, 6, 2), banded(2000, 8, 0.5, 3)]
    for A in mats:
        fs = [cp.ConcaveWorkModel(0.0, 0, 1), cp.ConcaveWorkModel(-0.7, 0, 1), cp.ConcaveWorkModel(0.5, 1, 1), cp.AffineWorkModel(0, 0, 0),
              cp.AffineWorkModel(-2, 3, 1), cp.AffineConnectivityModel(0, 3, 1, 3), cp.AffineConnectivityModel(-0.5, 0.0, 0.0, 1.0),
              cp.ConstrainedCost(cp.ConcaveWorkModel(0, 1, 0), w, 2), cp.ConstrainedCost(cp.ConcaveWorkModel(0, 1, 0), w, 4),
              cp.ConstrainedCost(cp.ConcaveWorkModel(0, 0, 1), w, 8), cp.ConstrainedCost(cp.AffineConnectivityModel(0, 3, 1, 3), cp.VertexCount(), 4)]
        for f in fs:
            if A.n >= 1:
                got = cp.pack_stripe(A, cp.ConcaveTotalChunker(f), backend=hip)
                want = cp.pack_stripe(A, cp.ConcaveTotalChunker(f), backend=orc)
                assert got == want, (A, "chunker", fs.index(f))
            for K in (1, 2, 3, 8):
                if A.n > 1000 and K > 3:
                    continue
                got = cp.partition_stripe(A, K, cp.ConcaveTotalSplitter(f), backend=hip)
                want = cp.partition_stripe(A, K, cp.ConcaveTotalSplitter(f), backend=orc)
                assert got == want, (A, K, "splitter", fs.index(f))


def test_power_work_model_on_device(hip, orc):
    """ConvexWorkModel (x^0.8, pow() within a few ulp of the host's) and ConcaveWorkModel (x*x, exact) through every
    method that takes a model: values within 1e-12 relative, and the exact DP reaches the oracle's optimum."""
    A = suitesparse_shaped(2000, 6, 4)
    for f in (cp.ConvexWorkModel(0.0, 0, 1), cp.ConvexWorkModel(-0.7, 1, 1), cp.ConcaveWorkModel(0.0, 0, 1)):
        j = np.array([1, 1, 5, 100, 1500], dtype=np.int64); jp = np.array([1, 2001, 900, 101, 2001], dtype=np.int64)
        a = cp.oracle_stripe(cp.RandomHint(), f, A, backend=hip)(j, jp)
        b = cp.oracle_stripe(cp.RandomHint(), f, A, backend=orc)(j, jp)
        assert np.allclose(a, b, rtol=1e-12, atol=0)
        for K in (2, 5):
            got = cp.partition_stripe(A, K, cp.DynamicTotalSplitter(f), backend=hip)
            want = cp.partition_stripe(A, K, cp.DynamicTotalSplitter(f), backend=orc)
            va, vb = cp.total_value(A, got, f, backend=orc), cp.total_value(A, want, f, backend=orc)
            assert abs(va - vb) <= 1e-12 * max(1.0, abs(vb))
        got = cp.pack_stripe(A, cp.ConvexTotalChunker(f), backend=hip) if f.gamma < 1 else cp.pack_stripe(A, cp.ConcaveTotalChunker(f), backend=hip)
        ref = cp.pack_stripe(A, cp.DynamicTotalChunker(f), backend=orc)
        va, vb = cp.total_value(A, got, f, backend=orc), cp.total_value(A, ref, f, backend=orc)
        assert abs(va - vb) <= 1e-9 * max(1.0, abs(vb))


def test_adjointpattern_on_device(hip, orc):
    """cp_adjoint / cp_csr_download: same arrays as the reference's counting sort; the adjoint keeps its device handle and can
    be partitioned right away (the alternating 2-D callers do exactly that)."""
    rng = np.random.default_rng(61)
    for A in _mats(34) + [suitesparse_shaped(20000, 8, 6), sprand(300, 5000, 0.01, rng), sprand(5000, 300, 0.01, rng)]:
        T = cp.adjointpattern(A, backend=hip)
        W = cp.adjointpattern(A, backend=orc)
        assert T.shape == W.shape and np.array_equal(T.colptr, W.colptr) and np.array_equal(T.rowval, W.rowval)
        f = cp.AffineConnectivityModel(0, 3, 1, 3)
        K = 3
        got = cp.partition_stripe(T, K, cp.DynamicTotalSplitter(f), backend=hip)
        want = cp.partition_stripe(W, K, cp.DynamicTotalSplitter(f), backend=orc)
        assert got == want
