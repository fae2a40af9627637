"""GPU parity: [Flip]BisectCostBottleneckSplitter (one-launch device kernel) vs the CPU oracle; bit-exact
split vectors including the Float64 bisection path."""
import numpy as np
import pytest

from util import cp, sprand, golden_matrices, suitesparse_shaped

pytestmark = pytest.mark.gpu


def test_bisect_cost_matches_oracle(hip, orc):
    rng = np.random.default_rng(12)
    mats = [sprand(m, n, 0.3, rng) for m in (1, 3, 8) for n in (1, 2, 3, 8, 40)] + list(golden_matrices().values())
    mats += [suitesparse_shaped(20000, 8, 6)]
    for A in mats:
        for K in (1, 2, 3, 8, 32):
            for f in (cp.AffineWorkModel(0, 10, 1), cp.AffineConnectivityModel(0, 10, 1, 100), cp.AffineConnectivityModel(0, 3, 1, 3),
                      cp.AffineConnectivityModel(0.5, 0.25, 1.0, 3.0), cp.AffineWorkModel(2.5, 0.5, 1.25)):
                for eps in (0.1, 0.01, 0.001):
                    got = cp.partition_stripe(A, K, cp.BisectCostBottleneckSplitter(f, eps), backend=hip)
                    want = cp.partition_stripe(A, K, cp.BisectCostBottleneckSplitter(f, eps), backend=orc)
                    assert got == want, (A, K, f.kind, f.dtype, eps)
            # decreasing cost for the Flip variant: alpha large, negative betas (bound_stripe: c_hi = alpha)
            f = cp.AffineWorkModel(1 + A.nnz + 3 * A.n, -3, -1)
            for eps in (0.1, 0.001):
                got = cp.partition_stripe(A, K, cp.FlipBisectCostBottleneckSplitter(f, eps), backend=hip)
                want = cp.partition_stripe(A, K, cp.FlipBisectCostBottleneckSplitter(f, eps), backend=orc)
                assert got == want, (A, K, "flip", eps)


def test_bisect_within_eps_of_optimum(hip):
    """test_Partitioners.jl:112: bottleneck <= (1+eps) * optimal bottleneck (optimal from the device DP)."""
    A = suitesparse_shaped(3000, 6, 2)
    for f in (cp.AffineWorkModel(0, 10, 1), cp.AffineConnectivityModel(0, 10, 1, 100)):
        for K in (2, 7):
            opt = cp.partition_stripe(A, K, cp.DynamicBottleneckSplitter(f), backend=hip)
            c = cp.bottleneck_value(A, opt, f, backend=hip)
            for eps in (0.1, 0.01):
                got = cp.partition_stripe(A, K, cp.BisectCostBottleneckSplitter(f, eps), backend=hip)
                assert cp.bottleneck_value(A, got, f, backend=hip) <= c * (1 + eps)


def test_cost_bisection_that_cannot_terminate_is_reported(hip, orc):
    """Non-positive cost bounds: `while c_lo * (1 + eps) < c_hi` never becomes false and the reference spins forever
    (BisectCostBottleneckSplitter.jl:41, LazyBisectCostBottleneckSplitter.jl:249).  A kernel must not: both the device
    and the oracle report a violated precondition once a probe moves no bound."""
    A = suitesparse_shaped(300, 4, 1)
    for b in (hip, orc):
        for meth in (cp.BisectCostBottleneckSplitter(cp.AffineWorkModel(-3, 0, 0), 0.01),
                     cp.BisectCostBottleneckSplitter(cp.AffineConnectivityModel(-50, 0, 0, 0), 0.01),
                     cp.LazyBisectCostBottleneckSplitter(cp.AffineConnectivityModel(-50, 0, 0, 0), 0.01)):
            with pytest.raises(AssertionError):
                cp.partition_stripe(A, 4, meth, backend=b)


def test_bisect_cost_batch_equals_the_loop_and_the_oracle(hip, orc):
    """cp_partition_bisect_cost_batch: B requests (K, model, eps, flip) on one pattern in ONE launch, one wave each, the counting
    structure built once.  Every split vector must be the single call's -- and the oracle's (BisectCostBottleneckSplitter.jl:6-63)."""
    from util import suitesparse_shaped
    A = suitesparse_shaped(20000, 8, 13)
    reqs = []
    for K in (2, 3, 7, 16, 32, 100):
        for eps in (0.1, 0.01, 0.001):
            reqs.append((K, cp.BisectCostBottleneckSplitter(cp.AffineWorkModel(0, 10, 1), eps)))
            reqs.append((K, cp.BisectCostBottleneckSplitter(cp.AffineConnectivityModel(0, 10, 1, 100), eps)))
            reqs.append((K, cp.BisectCostBottleneckSplitter(cp.AffineConnectivityModel(3, 0, 1, 7), eps)))
            reqs.append((K, cp.FlipBisectCostBottleneckSplitter(cp.AffineConnectivityModel(0, 10, 1, 100), eps)))
    got = cp.partition_stripe_batch(A, reqs, backend=hip)
    assert len(got) == len(reqs)
    for (K, m), g in zip(reqs, got):
        assert g == cp.partition_stripe(A, K, m, backend=hip), (K, type(m).__name__, m.eps)
    for (K, m), g in list(zip(reqs, got))[::5]:
        assert g == cp.partition_stripe(A, K, m, backend=orc), (K, type(m).__name__, m.eps)
    # Float64 batch; mixed element types are refused
    fr = [(K, cp.BisectCostBottleneckSplitter(cp.AffineConnectivityModel(0.5, 0.25, 0.0, 1.5), 0.01)) for K in (2, 5, 9)]
    for (K, m), g in zip(fr, cp.partition_stripe_batch(A, fr, backend=hip)):
        assert g == cp.partition_stripe(A, K, m, backend=orc)
    with pytest.raises(NotImplementedError):
        cp.partition_stripe_batch(A, reqs[:1] + fr[:1], backend=hip)
