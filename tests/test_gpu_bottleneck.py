"""GPU parity of the bottleneck DP (DynamicBottleneckSplitter / Chunker, g = max) on the valley-search path of
csrc/dp_bottleneck.hip: tables bit-exact against the oracle's literal sweep (tie-heavy integer costs, per-part alpha,
non-integral Float64), the general sweep as a second opinion, and at bench size the exact bottleneck value against
BisectIndexBottleneckSplitter (BisectIndexBottleneckSplitter.jl:5-83) -- a full-size cross-check with a non-trivial answer."""
import numpy as np
import pytest

from util import cp, sprand, golden_matrices, suitesparse_shaped, banded

pytestmark = pytest.mark.gpu

MODELS = [cp.AffineConnectivityModel(0, 0, 0, 1), cp.AffineConnectivityModel(0, 10, 1, 100), cp.AffineWorkModel(0, 10, 1), cp.AffineWorkModel(3, 0, 1),
          cp.AffineHyperedgeCutModel(0, 1, 0, 3, 2), cp.AffineHyperedgeCutModel(0, 0, 0, 1, 1), cp.AffineConnectivityModel(0.5, 0.25, 0.0, 1.5),
          cp.AffineConnectivityModel(0, 3, 1, 3, alpha_k=[5, 1, 9, 2, 7, 3, 8, 4]), cp.AffineConnectivityModel(0.0, 0.0, 0.0, 1.0),
          # non-integral hyperedge cut: OUTSIDE the valley class after rounding (tests/test_oracle_bottleneck.py) -> general sweep
          cp.AffineHyperedgeCutModel(0., 0., 0., 0.7, 0.1), cp.AffineHyperedgeCutModel(0.3, 0.1, 0., 0.3, 0.3),
          cp.AffineHyperedgeCutModel(0., 1., 0., 3., 2.)]


def mats():
    rng = np.random.default_rng(0xDEADBEEF)
    out = [sprand(m, n, p, rng) for (m, n, p) in [(1, 1, 0.5), (3, 2, 0.5), (5, 7, 0.4), (8, 16, 0.3), (10, 23, 0.2), (6, 33, 0.3), (20, 40, 0.1),
                                                  (9, 64, 0.2), (9, 65, 0.2), (40, 100, 0.05), (4, 8, 0.0), (3, 300, 0.5)]]
    out += list(golden_matrices().values())
    out += [suitesparse_shaped(1000, 6, 3), banded(777, 4, 0.5, 9), suitesparse_shaped(2100, 4, 8)]
    return out


@pytest.mark.parametrize("mi", range(len(MODELS)))
def test_bottleneck_tables_bit_exact(hip, orc, mi):
    mdl = MODELS[mi]
    for A in mats():
        for K in (1, 2, 5, 8):
            mm = mdl.marshal()
            rc1, p1, c1 = hip.dynamic_tables(A, K, 1, mm, None)
            rc2, p2, c2 = orc.dynamic_tables(A, K, 1, mm, None)
            assert rc1 == 0 and rc2 == 0, hip.last_error()
            assert np.array_equal(p1, p2), (A, K, mi)
            assert np.array_equal(c1, c2), (A, K, mi)
            for meth in (cp.DynamicBottleneckSplitter, cp.DynamicBottleneckChunker):
                if meth is cp.DynamicBottleneckChunker and mdl.alpha_k is not None:
                    continue
                got = cp.partition_stripe(A, K, meth(mdl), backend=hip)
                want = cp.partition_stripe(A, K, meth(mdl), backend=orc)
                assert got == want, (A, K, mi)


def test_chunk_sizes_and_general_sweep_agree(hip, orc):
    for A in (suitesparse_shaped(3000, 8, 11), suitesparse_shaped(3000, 30, 5)):
      for mdl in (MODELS[1], MODELS[4], MODELS[6]):
        K = 6
        mm = mdl.marshal()
        rc2, p2, c2 = orc.dynamic_tables(A, K, 1, mm, None)
        # the lane-per-chunk walk (bn_wave 0) at several chunk sizes, the wave-per-run walk (default) at several run lengths:
        # 2 .. 64 rows = one sub-run, 65 / 127 / 253 = sub-runs anchored at their predecessor's last row, 100000 = one wave for the layer
        # (bn_wave 2, the default: the crossings of a sub-run by binary search over windows of 64 columns -- Int64 costs; Float64 costs take the lockstep walk)
        for wave, ch in ((0, 1), (0, 3), (0, 7), (0, 16), (0, 128), (0, 100000), (1, 2), (1, 3), (1, 63), (1, 64), (1, 65), (1, 127), (1, 253), (1, 1000), (1, 100000),
                         (2, 2), (2, 3), (2, 63), (2, 64), (2, 65), (2, 127), (2, 253), (2, 1000), (2, 100000)):
            hip.set_option("bn_wave", wave); hip.set_option("bn_chunk" if not wave else "bn_run", ch)
            try:
                rc1, p1, c1 = hip.dynamic_tables(A, K, 1, mm, None)
            finally:
                hip.set_option("bn_chunk", 8); hip.set_option("bn_run", 253); hip.set_option("bn_wave", 2)
            assert rc1 == 0 and np.array_equal(p1, p2) and np.array_equal(c1, c2), (wave, ch)
        hip.set_option("force_brute", 1)
        try:
            rc1, p1, c1 = hip.dynamic_tables(A, K, 1, mm, None)
        finally:
            hip.set_option("force_brute", 0)
        assert np.array_equal(p1, p2) and np.array_equal(c1, c2)


def test_bottleneck_at_bench_size_equals_bisect_index(hip):
    """n = 10^7 / nnz = 10^8 is too much host memory for a numpy generator inside the suite: n = 2*10^6, K = 64 here (bench.py
    runs the 10^7 case).  The DP's bottleneck value must equal the exact BisectIndex optimum; both return non-trivial splits."""
    A = suitesparse_shaped(2_000_000, 8, 77)
    K = 64
    for mdl in (cp.AffineConnectivityModel(0, 10, 1, 100), cp.AffineWorkModel(0, 10, 1)):
        dp = cp.partition_stripe(A, K, cp.DynamicBottleneckSplitter(mdl), backend=hip)
        bi = cp.partition_stripe(A, K, cp.BisectIndexBottleneckSplitter(mdl), backend=hip)
        v_dp = cp.bottleneck_value(A, dp, mdl, backend=hip)
        v_bi = cp.bottleneck_value(A, bi, mdl, backend=hip)
        assert v_dp == v_bi
        assert len(set(dp.spl.tolist())) > K // 2


# ------------------------------------------------------------------ width-constrained bottleneck DP
# DynamicBottleneck{Splitter,Chunker}(ConstrainedCost(f, VertexCount(), w_max)) (DynamicSplitter.jl:206-314 with g = max): the
# searched-crossings walk with per-row candidate limits.  Int64 models take it; Float64 falls to the literal one-wave kernel.
CMODELS = [cp.AffineConnectivityModel(0, 0, 0, 1), cp.AffineConnectivityModel(0, 10, 1, 100), cp.AffineWorkModel(0, 10, 1),
           cp.AffineHyperedgeCutModel(0, 1, 0, 3, 2), cp.AffineConnectivityModel(0, 3, 1, 3, alpha_k=[5, 1, 9, 2, 7, 3, 8, 4])]


def _widths(n, K):
    return sorted({max(1, -(-n // K)), max(1, -(-3 * n // (2 * K))), max(1, n // 2), max(1, n - 1), n + 3, 1, 2, 3, 5, 8, 63, 64, 65})


@pytest.mark.parametrize("mi", range(len(CMODELS)))
def test_constrained_bottleneck_tables_bit_exact(hip, orc, mi):
    mdl = CMODELS[mi]
    nondeg = 0
    for A in mats():
        for K in (1, 2, 5, 8):
            for w in _widths(A.n, K):
                mm = mdl.marshal()
                rc2, lo2, hi2, p2, c2 = orc.dynamic_tables_constrained(A, K, 1, mm, None, cp.VertexCount().marshal(), w, float(w))
                rc1, lo1, hi1, p1, c1 = hip.dynamic_tables_constrained(A, K, mm, w, combine=1)
                assert rc1 == rc2, (A, K, w, hip.last_error())
                assert np.array_equal(lo1, lo2) and np.array_equal(hi1, hi2), (A, K, w)
                if rc2 == 0:
                    assert np.array_equal(p1, p2), (A, K, w, mi)
                    assert np.array_equal(c1, c2), (A, K, w, mi)
                for meth in (cp.DynamicBottleneckSplitter, cp.DynamicBottleneckChunker):
                    if meth is cp.DynamicBottleneckChunker and mdl.alpha_k is not None:
                        continue
                    f = cp.ConstrainedCost(mdl, cp.VertexCount(), w)
                    got = cp.partition_stripe(A, K, meth(f), backend=hip)
                    want = cp.partition_stripe(A, K, meth(f), backend=orc)
                    assert got == want, (A, K, w, mi, meth.__name__)
                nondeg += int(len(set(want.spl.tolist())) > 2)
    assert nondeg > 100


def test_constrained_bottleneck_run_lengths_and_weights(hip, orc):
    """sub-run lengths around the wave width, hinted / unhinted starts, and an AffineWorkModel width weight (test_Partitioners.jl:178-183)"""
    mats_ = [suitesparse_shaped(3000, 8, 1), banded(2500, 6, 0.5, 3), suitesparse_shaped(1025, 5, 7)]
    try:
        for run, slack in ((253, 64), (2, 64), (63, 0), (64, 3), (65, 64), (1000, 64), (100000, 64)):
            hip.set_option("bn_run", run); hip.set_option("bn_slack", slack)
            for A in mats_:
                for mdl in (CMODELS[1], CMODELS[3]):
                    for (K, w) in [(4, -(-3 * A.n // 8)), (7, A.n // 4), (16, A.n // 8), (3, A.n // 3 + 97), (3, 700)]:
                        mm = mdl.marshal()
                        rc2, lo2, hi2, p2, c2 = orc.dynamic_tables_constrained(A, K, 1, mm, None, cp.VertexCount().marshal(), w, float(w))
                        rc1, lo1, hi1, p1, c1 = hip.dynamic_tables_constrained(A, K, mm, w, combine=1)
                        assert rc1 == rc2, hip.last_error()              # ((3, 700) is infeasible on the larger two: both say so)
                        if rc2 == 0:
                            assert np.array_equal(p1, p2) and np.array_equal(c1, c2), (A, K, w, run, slack)
    finally:
        hip.set_option("bn_run", 253); hip.set_option("bn_slack", 64)
    A = mats_[0]
    f = cp.ConstrainedCost(CMODELS[1], cp.AffineWorkModel(0, 1, 0), 800)
    got = cp.partition_stripe(A, 5, cp.DynamicBottleneckSplitter(f), backend=hip)
    want = cp.partition_stripe(A, 5, cp.DynamicBottleneckSplitter(f), backend=orc)
    assert got == want


def test_constrained_bottleneck_larger(hip, orc):
    """n = 1500: the windowed valley search against the one-wave literal kernel (force_brute; Theta(n w) oracle steps on one wave:
    40 s at n = 6000); n = 20 000: against the CPU oracle's literal DP"""
    A = suitesparse_shaped(1500, 8, 5)
    for (K, w) in [(8, 282), (5, 400)]:
        f = cp.ConstrainedCost(CMODELS[1], cp.VertexCount(), w)
        got = cp.partition_stripe(A, K, cp.DynamicBottleneckSplitter(f), backend=hip)
        hip.set_option("force_brute", 1)
        try:
            want = cp.partition_stripe(A, K, cp.DynamicBottleneckSplitter(f), backend=hip)
        finally:
            hip.set_option("force_brute", 0)
        assert got == want and len(set(got.spl.tolist())) == K + 1
    A = suitesparse_shaped(20000, 8, 6)
    free = cp.partition_stripe(A, 6, cp.DynamicBottleneckSplitter(CMODELS[3]), backend=orc)
    for w in (3340, 3360, 4000):                                   # 3340: six more than n / K -- the constraint decides the answer
        f = cp.ConstrainedCost(CMODELS[3], cp.VertexCount(), w)
        got = cp.partition_stripe(A, 6, cp.DynamicBottleneckSplitter(f), backend=hip)
        assert got == cp.partition_stripe(A, 6, cp.DynamicBottleneckSplitter(f), backend=orc) and len(set(got.spl.tolist())) == 7
        assert (got == free) == (w == 4000)


# pin-weighted budgets (and any AffineWorkModel weight with b_v, b_p >= 0): the window's left end j0(j') is an array, not j' - w
WEIGHTS = [cp.AffineWorkModel(0, 0, 1), cp.AffineWorkModel(2, 3, 1), cp.AffineWorkModel(0.5, 0.0, 0.25), cp.AffineWorkModel(0, 1, 2)]


@pytest.mark.parametrize("wi", range(len(WEIGHTS)))
def test_pin_weighted_bottleneck_tables_bit_exact(hip, orc, wi):
    wgt = WEIGHTS[wi]
    a0, bv, bp = wgt._params()
    nondeg = infeasible = 0
    for A in mats():
        for K in (1, 2, 5, 8):
            full = a0 + bv * A.n + bp * A.nnz                       # the weight of the whole matrix
            for frac in (1.0 / K, 1.5 / K, 0.5, 1.0, 0.0):
                wmax = a0 + (full - a0) * frac + (1 if frac else 0)
                wmax = int(wmax) if wgt.dtype == cp.models.CP_I64 else float(wmax)
                for mdl in (CMODELS[1], CMODELS[3]):
                    mm = mdl.marshal(); wm = wgt.marshal()
                    rc2, lo2, hi2, p2, c2 = orc.dynamic_tables_constrained(A, K, 1, mm, None, wm, int(wmax), float(wmax))
                    rc1, lo1, hi1, p1, c1 = hip.dynamic_tables_constrained(A, K, mm, wmax, combine=1, wm=wgt.marshal())
                    assert rc1 == rc2, (A, K, wmax, hip.last_error())
                    assert np.array_equal(lo1, lo2) and np.array_equal(hi1, hi2), (A, K, wmax)
                    infeasible += rc2 != 0
                    if rc2 == 0:
                        assert np.array_equal(p1, p2), (A, K, wmax, wi)
                        assert np.array_equal(c1, c2), (A, K, wmax, wi)
                    f = cp.ConstrainedCost(mdl, wgt, wmax)
                    for meth in (cp.DynamicBottleneckSplitter, cp.DynamicBottleneckChunker):
                        got = cp.partition_stripe(A, K, meth(f), backend=hip)
                        want = cp.partition_stripe(A, K, meth(f), backend=orc)
                        assert got == want, (A, K, wmax, wi, meth.__name__)
                    nondeg += int(len(set(want.spl.tolist())) > 2)
    assert nondeg > 100 and infeasible > 10


def test_pin_weighted_bottleneck_larger(hip, orc):
    """a budget of 1.3 x the mean pins per part: the j0-array valley search against the one-wave literal kernel (n = 1500) and the
    CPU oracle (n = 20 000)"""
    A = suitesparse_shaped(1500, 8, 5)
    K = 6
    f = cp.ConstrainedCost(CMODELS[1], cp.AffineWorkModel(0, 0, 1), int(1.3 * A.nnz / K))
    got = cp.partition_stripe(A, K, cp.DynamicBottleneckSplitter(f), backend=hip)
    hip.set_option("force_brute", 1)
    try:
        want = cp.partition_stripe(A, K, cp.DynamicBottleneckSplitter(f), backend=hip)
    finally:
        hip.set_option("force_brute", 0)
    assert got == want and len(set(got.spl.tolist())) == K + 1
    A = suitesparse_shaped(20000, 8, 6)
    free = cp.partition_stripe(A, K, cp.DynamicBottleneckSplitter(CMODELS[3]), backend=orc)
    for wgt, budget in ((cp.AffineWorkModel(0, 2, 1), int(1.005 * (2 * A.n + A.nnz) / K)), (cp.AffineWorkModel(0, 0, 1), int(1.002 * A.nnz / K)),
                        (cp.AffineWorkModel(0, 0, 1), int(1.3 * A.nnz / K))):
        f = cp.ConstrainedCost(CMODELS[3], wgt, budget)
        got = cp.partition_stripe(A, K, cp.DynamicBottleneckSplitter(f), backend=hip)
        assert got == cp.partition_stripe(A, K, cp.DynamicBottleneckSplitter(f), backend=orc) and len(set(got.spl.tolist())) == K + 1
        assert (got == free) == (budget == int(1.3 * A.nnz / K))        # the tight budgets decide the answer
