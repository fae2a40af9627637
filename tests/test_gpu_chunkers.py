"""GPU parity of the sequential partitioners (csrc/seq.hip) against the CPU oracle: bit-exact split
vectors for pack_stripe(DynamicTotalChunker / ConvexTotalChunker), ConvexTotalSplitter, the
ConstrainedCost DP variants, and the block-cost oracle."""
import numpy as np
import pytest

from util import cp, sprand, golden_matrices, suitesparse_shaped, banded

pytestmark = pytest.mark.gpu


def mats(seed):
    rng = np.random.default_rng(seed)
    out = [sprand(m, n, p, rng) for (m, n, p) in [(1, 1, 0.5), (2, 3, 0.5), (4, 8, 0.3), (8, 8, 0.3), (8, 16, 0.3), (12, 31, 0.15),
                                                  (20, 40, 0.1), (30, 65, 0.08)]]
    g = golden_matrices()
    out += [g["LPnetlib/lpi_itest6"], g["Pajek/GD99_c"], g["LPnetlib/lp_blend"]]
    out += [suitesparse_shaped(200, 4, 7), banded(150, 3, 0.5, 3)]
    return out


FS = [cp.AffineConnectivityModel(0, 3, 1, 3), cp.AffineConnectivityModel(0, 0, 0, 1), cp.AffineConnectivityModel(-0.5, 0.0, 0.0, 1.0),
      cp.AffineWorkModel(0, 0, 0), cp.AffineWorkModel(0, 10, 1), cp.AffineHyperedgeCutModel(0, 1, 1, 1, 3),
      cp.ColumnBlockComponentCostModel(3, lambda w: 1 + w)]


def test_pack_dynamic(hip, orc):
    for A in mats(1):
        for f in FS:
            variants = [f] + [cp.ConstrainedCost(f, cp.VertexCount(), w) for w in (2, 4, 8)] + [cp.ConstrainedCost(f, cp.AffineWorkModel(0, 1, 0), 4)]
            for fc in variants:
                got = cp.pack_stripe(A, cp.DynamicTotalChunker(fc), backend=hip)
                want = cp.pack_stripe(A, cp.DynamicTotalChunker(fc), backend=orc)
                assert got == want, (A, f.kind)
                assert cp.total_value(A, got, f, backend=hip) == cp.total_value(A, want, f, backend=orc)


def test_pack_convex(hip, orc):
    for A in mats(2):
        for f in FS:
            variants = [f] + [cp.ConstrainedCost(f, cp.VertexCount(), w) for w in (2, 4, 8)] + [cp.ConstrainedCost(f, cp.AffineWorkModel(0, 1, 0), 4)]
            for fc in variants:
                got = cp.pack_stripe(A, cp.ConvexTotalChunker(fc), backend=hip)
                want = cp.pack_stripe(A, cp.ConvexTotalChunker(fc), backend=orc)
                assert got == want, (A, f.kind)


def test_partition_convex_and_constrained_dp(hip, orc):
    for A in mats(3):
        for K in (1, 2, 3, 4, 8):
            for f in FS[:5]:
                got = cp.partition_stripe(A, K, cp.ConvexTotalSplitter(f), backend=hip)
                want = cp.partition_stripe(A, K, cp.ConvexTotalSplitter(f), backend=orc)
                assert got == want, (A, K, f.kind)
                for w_max in (2, 4, 8):
                    fc = cp.ConstrainedCost(f, cp.AffineWorkModel(0, 1, 0), w_max)
                    for meth in (cp.ConvexTotalSplitter(fc), cp.DynamicTotalSplitter(fc), cp.DynamicBottleneckSplitter(fc),
                                 cp.DynamicTotalChunker(fc), cp.DynamicBottleneckChunker(fc)):
                        got = cp.partition_stripe(A, K, meth, backend=hip)
                        want = cp.partition_stripe(A, K, meth, backend=orc)
                        assert got == want, (A, K, f.kind, w_max, type(meth).__name__)


def test_block_costs(hip, orc):
    """BlockComponentCostStepOracle on the device: total_value and width-limited chunking (test_Costs.jl:106-122,
    test_Partitioners.jl:225-248)."""
    rng = np.random.default_rng(6)
    for m in (3, 8, 17, 30):
        A = sprand(m, m, 0.2, rng)
        for u in (1, 2, 4):
            Pi = cp.pack_stripe(A, cp.EquiChunker(u))
            for mdl in (cp.BlockComponentCostModel(0, 0, (2, lambda x: x), (2, lambda x: 2 * x)),
                        cp.BlockComponentCostModel(lambda x: x, lambda x: 3 * x, (10, lambda x: x), (2, lambda x: 2 * x)),
                        cp.BlockComponentCostModel(1, 3, (1,), (1,))):
                for w in (1, 2, 4):
                    Phi = cp.pack_stripe(A, cp.EquiChunker(w))
                    assert cp.total_value(A, Phi, mdl, Pi, backend=hip) == cp.total_value(A, Phi, mdl, Pi, backend=orc)
                f = cp.ConstrainedCost(mdl, cp.VertexCount(), 4)
                got = cp.pack_stripe(A, cp.DynamicTotalChunker(f), Pi, backend=hip)
                want = cp.pack_stripe(A, cp.DynamicTotalChunker(f), Pi, backend=orc)
                assert got == want
                assert np.all(np.diff(got.spl) <= 4)


def test_pack_dynamic_scan_path(hip, orc):
    """Width-windowed DynamicTotalChunker takes the parallel (min,+) scan (csrc/chunk_scan.hip): same split vector as the
    oracle, and as the one-wave literal kernel (cp_set_option("force_brute")), across window widths, block boundaries
    of the scan (n around multiples of 64) and models with negative costs / frequent ties."""
    fs = [cp.AffineConnectivityModel(0, 3, 1, 3), cp.AffineConnectivityModel(0, 0, 0, 1), cp.AffineConnectivityModel(-7, 0, 0, 1),
          cp.AffineWorkModel(0, 0, 0), cp.AffineWorkModel(-3, 1, 0), cp.AffineHyperedgeCutModel(0, 1, 1, 1, 3),
          cp.ColumnBlockComponentCostModel(3, lambda w: 1 + w), cp.AffineConnectivityModel(0.0, 0.0, 0.0, 1.0),
          cp.AffineConnectivityModel(-0.5, 0.0, 0.0, 1.0)]
    for n in (1, 2, 63, 64, 65, 129, 1000, 4097, 20000):
        A = banded(n, 5, 0.5, n) if n % 2 else suitesparse_shaped(n, 4, n)
        for f in fs:
            for w in (1, 2, 3, 5, 8, 9, 16, 17):
                if n > 5000 and w not in (3, 8, 16):
                    continue
                fc = cp.ConstrainedCost(f, cp.VertexCount(), w)
                got = cp.pack_stripe(A, cp.DynamicTotalChunker(fc), backend=hip)
                if n <= 5000:
                    want = cp.pack_stripe(A, cp.DynamicTotalChunker(fc), backend=orc)
                    assert got == want, (n, w, fs.index(f))
                hip.set_option("force_brute", 1)
                try:
                    lit = cp.pack_stripe(A, cp.DynamicTotalChunker(fc), backend=hip)
                finally:
                    hip.set_option("force_brute", 0)
                assert got == lit, (n, w, fs.index(f), "scan vs literal kernel")
                assert np.all(np.diff(got.spl) <= w)


def test_convex_chunker_on_a_non_convex_cost_at_scale(hip, orc):
    """Config-4 shape: ConvexTotalChunker(ConstrainedCost(col_block_model, VertexCount(), 8)) on a banded matrix.  The cost is
    not convex, the candidate stack goes stale from n ~ 1e5 on and the reference evaluates the closure at NEGATIVE widths
    (ConvexTotalChunker.jl:76,99): the tables cover them (cp_component_t.lo) and the device must follow the oracle exactly."""
    A = banded(120000, 16, 0.5, 4)
    f = cp.ConstrainedCost(cp.ColumnBlockComponentCostModel(3, lambda w: 1 + w), cp.VertexCount(), 8)
    got = cp.pack_stripe(A, cp.ConvexTotalChunker(f), backend=hip)
    want = cp.pack_stripe(A, cp.ConvexTotalChunker(f), backend=orc)
    assert got == want
    assert np.all(np.diff(got.spl) <= 8)
    got = cp.pack_stripe(A, cp.DynamicTotalChunker(f), backend=hip)
    want = cp.pack_stripe(A, cp.DynamicTotalChunker(f), backend=orc)
    assert got == want


def test_pack_convex_batch_equals_the_loop_and_the_oracle(hip, orc):
    """cp_pack_convex_batch: B ConvexTotalChunker requests (model constants x width limits) on one pattern in ONE launch, one wave each,
    sharing the net counter and the window table of net counts.  Every chunk vector must be the single call's -- and the oracle's."""
    from util import banded, suitesparse_shaped
    for A in (banded(3000, 16, 0.5, 4), suitesparse_shaped(2500, 6, 9)):
        meths = []
        for w in (1, 2, 4, 8, 12, 15):
            for a in (1, 3, 10):
                meths.append(cp.ConvexTotalChunker(cp.ConstrainedCost(cp.ColumnBlockComponentCostModel(a, lambda x, a=a: a + x), cp.VertexCount(), w)))
            meths.append(cp.ConvexTotalChunker(cp.ConstrainedCost(cp.AffineConnectivityModel(2, 3, 1, 5), cp.VertexCount(), w)))
            meths.append(cp.ConvexTotalChunker(cp.ConstrainedCost(cp.AffineWorkModel(7, 1, 2), cp.VertexCount(), w)))
        got = cp.pack_stripe_batch(A, meths, backend=hip)
        assert len(got) == len(meths)
        for m, g in zip(meths, got):
            one = cp.pack_stripe(A, m, backend=hip)
            assert g == one, (A, type(M_ := cp.models.split_constraint(m.f)[0]).__name__, cp.models.split_constraint(m.f)[2])
            assert int(np.diff(g.spl).max()) <= cp.models.split_constraint(m.f)[2]
        for m, g in list(zip(meths, got))[::4]:
            assert g == cp.pack_stripe(A, m, backend=orc)
