"""GPU parity of the width-constrained DP, DynamicTotal{Splitter,Chunker}(ConstrainedCost(f, VertexCount(), w_max))
(DynamicSplitter.jl:206-314) on the O(K n log^2 n) windowed path of csrc/dp_total.hip.  Unlike the unconstrained DP its
answer is NOT closed-form (the split vectors are non-degenerate), so `spl` itself discriminates:
  * every in-window cst / ptr cell and the windows j'_lo / j'_hi against the oracle's tables (cp_dynamic_tables_constrained);
  * split vectors against the oracle, both loop orders, all driver options;
  * the windowed layer kernel driven directly with injected previous rows against brute force (tests/brute.py);
  * against the one-wave literal kernel (force_brute) at a size the oracle does not reach.
"""
import numpy as np
import pytest
import torch

import brute
from util import cp, sprand, golden_matrices, suitesparse_shaped, banded

pytestmark = pytest.mark.gpu

MODELS = [cp.AffineConnectivityModel(0, 0, 0, 1), cp.AffineConnectivityModel(0, 10, 1, 100), cp.AffineWorkModel(0, 10, 1),
          cp.AffineHyperedgeCutModel(0, 0, 0, 0, 1), cp.AffineHyperedgeCutModel(0, 2, 1, 1, 3),
          cp.AffineConnectivityModel(0.0, 0.0, 0.0, 1.0), cp.AffineConnectivityModel(0, 3, 1, 3, alpha_k=[5, 1, 9, 2, 7, 3, 8, 4])]


def mats():
    rng = np.random.default_rng(0xDEADBEEF)
    out = [sprand(m, n, p, rng) for (m, n, p) in [(3, 2, 0.5), (5, 7, 0.4), (8, 16, 0.3), (10, 23, 0.2), (6, 33, 0.3), (20, 40, 0.1),
                                                  (9, 64, 0.2), (9, 65, 0.2), (9, 63, 0.2), (40, 100, 0.05), (4, 8, 0.0)]]
    out += list(golden_matrices().values())
    out += [suitesparse_shaped(1000, 6, 3), banded(777, 4, 0.5, 9)]
    return out


def widths(n, K):
    ws = {max(1, -(-n // K)), max(1, -(-3 * n // (2 * K))), max(1, n // 2), max(1, n - 1), n + 3, 1, 2, 3, 5, 8}
    return sorted(ws)


@pytest.mark.parametrize("mi", range(len(MODELS)))
def test_constrained_tables_bit_exact(hip, orc, mi):
    mdl = MODELS[mi]
    nondeg = 0
    for A in mats():
        for K in (1, 2, 5, 8):
            for w in widths(A.n, K):
                mm = mdl.marshal()
                wm = cp.VertexCount().marshal()
                rc2, lo2, hi2, p2, c2 = orc.dynamic_tables_constrained(A, K, 0, mm, None, wm, w, float(w))
                rc1, lo1, hi1, p1, c1 = hip.dynamic_tables_constrained(A, K, mm, w)
                assert rc1 == rc2, (A, K, w, hip.last_error())
                assert np.array_equal(lo1, lo2) and np.array_equal(hi1, hi2), (A, K, w)
                if rc2 == 0:
                    assert np.array_equal(p1, p2), (A, K, w, mi)
                    assert np.array_equal(c1, c2), (A, K, w, mi)
                for meth in (cp.DynamicTotalSplitter, cp.DynamicTotalChunker):
                    f = cp.ConstrainedCost(mdl, cp.VertexCount(), w)
                    got = cp.partition_stripe(A, K, meth(f), backend=hip)
                    want = cp.partition_stripe(A, K, meth(f), backend=orc)
                    assert got == want, (A, K, w, mi, meth.__name__)
                nondeg += int(len(set(want.spl.tolist())) > 2)
    assert nondeg > 100


OPTIONS = [{}, {"nospec": 1}, {"gap_tau": -1}, {"gap_tau": 8, "gap_min": 8}, {"dbg": 64}, {"dbg": 512, "gap_tau": 7, "gap_min": 8},
           {"short_t": 0, "short_e": 0}, {"own_min": 1000}, {"rpass_small_tau": 6}, {"rpass_ch": 16}, {"rpass_small_tau": -1, "rpass_ch": 16}, {"rpass_cap": 1}, {"dbg": 524288}, {"dbg": 1048576}, {"dbg": 2097152}, {"dbg": 8388608}, {"ra_cache": 0}, {"force_max": 1000000}, {"dbg": 16384}, {"setup_bs": 128}, {"rpass_small_tau": 9, "rpass_cap": 30}, {"dbg": 1024}, {"dbg": 2048},
           # without the leaf pass (csrc/dp_leaf.inc): the rounds tau < 6 as divide-and-conquer rounds
           {"leaf": 0}, {"leaf": 0, "gap_tau": 8, "gap_min": 8}, {"gap_nr": 1}, {"gap_nr": 1, "gap_tau": 8, "gap_min": 8}, {"dbg": 33554432}]
DEFAULTS = {"leaf": 1, "ra_cache": 1, "nospec": 0, "gap_tau": 6, "gap_min": 64, "dbg": 0, "rpass_small_tau": 4, "rpass_ch": 256, "rpass_cap": 200, "force_max": 1024, "setup_bs": 1024, "short_t": 8, "short_e": 64, "own_min": 64, "gap_nr": 2}


@pytest.mark.parametrize("oi", range(len(OPTIONS)))
def test_constrained_every_layer_option(hip, orc, oi):
    mats_ = [suitesparse_shaped(3000, 8, 1), banded(2500, 6, 0.5, 3), suitesparse_shaped(1025, 5, 7)]
    try:
        for k, v in OPTIONS[oi].items():
            assert hip.set_option(k, v) == 0
        for A in mats_:
            for mdl in (MODELS[0], MODELS[1], MODELS[4]):
                for (K, w) in [(4, -(-3 * A.n // 8)), (7, A.n // 4), (16, A.n // 8), (3, A.n // 3 + 97), (3, 700)]:
                    mm = mdl.marshal()
                    rc2, lo2, hi2, p2, c2 = orc.dynamic_tables_constrained(A, K, 0, mm, None, cp.VertexCount().marshal(), w, float(w))
                    rc1, lo1, hi1, p1, c1 = hip.dynamic_tables_constrained(A, K, mm, w)
                    assert rc1 == rc2 and rc1 in (0, 2), hip.last_error()          # 2: infeasible windows (K w_max < n)
                    assert rc1 == 2 or (np.array_equal(p1, p2) and np.array_equal(c1, c2)), (OPTIONS[oi], A, K, w)
    finally:
        for k, v in DEFAULTS.items():
            hip.set_option(k, v)


def test_windowed_layer_with_injected_rows(hip):
    """the layer kernel itself: min over max(0, r - w) <= p <= r of W[p] + f(p, r) for arbitrary rows W (rows that are not a
    DP layer move the winners around inside the window), against brute force"""
    rng = np.random.default_rng(17)
    moved = 0
    for A in [sprand(9, 65, 0.2, rng), sprand(40, 100, 0.05, rng), suitesparse_shaped(1500, 6, 5), banded(1200, 5, 0.5, 2)]:
        n = A.n
        NT, ST = brute.net_table(A), brute.selfnet_table(A)
        for mdl in (MODELS[1], MODELS[4]):
            F = brute.cost_table(A, mdl, 2, NT, ST)
            scale = int(abs(F).max()) + 1
            # (63 .. 129: around the smallest windows the leaf pass takes -- s = 6 -- and its inner mirrored candidates [r - w, 64 g + 63 - w))
            for w in sorted({1, 2, 3, 7, 33, 63, 64, 65, 100, 127, 128, 129, n // 5, n // 2, n}):
                rows = [rng.integers(0, scale + 1, n + 1), np.sort(rng.integers(0, scale + 1, n + 1)), rng.integers(0, 3, n + 1),
                        np.where(rng.random(n + 1) < 0.02, 0, scale * 8).astype(np.int64)]
                for W in rows:
                    cst, ptr = hip.windowed_layer(A, mdl.marshal(), W.astype(np.int64), w)
                    lo = np.maximum(0, np.arange(n + 1) - w)
                    cb, pb = brute.layer(W, F, lo=lo)
                    assert np.array_equal(ptr, pb), (A, w)
                    assert np.array_equal(cst, cb), (A, w)
                    moved += int(np.sum(pb != np.arange(n + 1)))
    assert moved > 10000


def test_windowed_path_equals_literal_kernel_beyond_the_oracle(hip):
    """n = 2e5, K = 8, w = ceil(1.5 n / K): the O(K n log^2 n) path against the one-wave literal kernel (force_brute)"""
    A = suitesparse_shaped(200_000, 8, 42)
    K = 8
    w = -(-3 * A.n // (2 * K))
    for mdl in (MODELS[0], MODELS[1]):
        f = cp.ConstrainedCost(mdl, cp.VertexCount(), w)
        fast = cp.partition_stripe(A, K, cp.DynamicTotalSplitter(f), backend=hip)
        assert len(set(fast.spl.tolist())) >= -(-A.n // w) + 1   # a non-degenerate answer: at least ceil(n / w) non-empty parts
        assert np.all(np.diff(fast.spl) <= w)
        # the literal kernel is Theta(sum of window^2) on one wave: run it on a prefix-sized problem only
    B = suitesparse_shaped(1500, 8, 43)          # (the literal kernel is one wave doing Theta(n^2) oracle steps: 40 s at n = 6000)
    for mdl in (MODELS[1], MODELS[4]):
        for (K, w) in [(8, 282), (5, 400)]:
            f = cp.ConstrainedCost(mdl, cp.VertexCount(), w)
            fast = cp.partition_stripe(B, K, cp.DynamicTotalSplitter(f), backend=hip)
            hip.set_option("force_brute", 1)
            try:
                lit = cp.partition_stripe(B, K, cp.DynamicTotalSplitter(f), backend=hip)
            finally:
                hip.set_option("force_brute", 0)
            assert fast == lit


def test_round_a_caches_change_no_table_cell_at_3e5(hip):
    """Beyond the oracle's reach (s = 16: heads with blocks of up to 2^15 candidates, merged from tile partials): the complete
    cst / ptr tables of the width-constrained DP with round A from the cached counts (default) equal those of the generic round A
    (dbg 1048576: no cache at all; dbg 2097152: standard heads cached, mirrored ones streamed), connectivity and hyperedge costs."""
    A = suitesparse_shaped(300000, 8, 3)
    for mdl, K, w in ((cp.AffineConnectivityModel(0, 10, 1, 100), 6, 75001), (cp.AffineHyperedgeCutModel(0, 2, 1, 1, 3), 5, 131072),
                      (cp.AffineConnectivityModel(0.0, 0.0, 0.0, 1.0), 9, 40000)):
        mm = mdl.marshal()
        rc0, lo0, hi0, p0, c0 = hip.dynamic_tables_constrained(A, K, mm, w)
        assert rc0 == 0, hip.last_error()
        for dbg in (1048576, 2097152, 8388608):
            hip.set_option("dbg", dbg)
            try:
                rc1, lo1, hi1, p1, c1 = hip.dynamic_tables_constrained(A, K, mm, w)
            finally:
                hip.set_option("dbg", 0)
            assert rc1 == 0 and np.array_equal(lo0, lo1) and np.array_equal(hi0, hi1)
            assert np.array_equal(p0, p1) and np.array_equal(c0, c1), (K, w, dbg)


def test_width_weights_take_the_windowed_path(hip, orc):
    """The reference's own tests constrain with AffineWorkModel(0, 1, 0) -- the width under another name
    (test/test_Partitioners.jl:178-183, 256-261).  Any weight alpha + c * width (c > 0, no pin term) bounds the parts by a number of
    columns: it must give the oracle's (literal, Theta(sum window^2)) answer AND run on the windowed O(K n log^2 n) path -- the leaf
    pass only exists there (the one-wave literal kernel never launches it)."""
    net = MODELS[1]
    weights = [(cp.AffineWorkModel(0, 1, 0), lambda w: w), (cp.AffineWorkModel(0, 3, 0), lambda w: 3 * w + 2), (cp.AffineWorkModel(5, 2, 0), lambda w: 2 * w + 5),
               (cp.AffineWorkModel(0.0, 0.5, 0.0), lambda w: 0.5 * w + 0.25), (cp.AffineWorkModel(0.25, 0.1, 0.0), lambda w: 0.1 * w + 0.25)]
    nondeg = 0
    for A in [suitesparse_shaped(3000, 8, 1), banded(2500, 6, 0.5, 3), golden_matrices()["HB/can_292"]]:
        n = A.n
        for K in (3, 8):
            for w in sorted({-(-3 * n // (2 * K)), n // 2, 100}):
                for wm, wmax_of in weights:
                    f = cp.ConstrainedCost(net, wm, wmax_of(w))
                    for meth in (cp.DynamicTotalSplitter, cp.DynamicTotalChunker):
                        hip.prof_reset(); hip.prof_enable(True)
                        try:
                            got = cp.partition_stripe(A, K, meth(f), backend=hip)
                        finally:
                            hip.prof_enable(False)
                        want = cp.partition_stripe(A, K, meth(f), backend=orc)
                        assert got == want, (A, K, w, wm.beta_vertex, meth.__name__)
                        if w >= 64 and got.spl[1] != 1:          # (feasible, and the window spans a leaf group)
                            assert hip.prof_get().get("dp_leaf", {"launches": 0})["launches"] > 0, "not on the windowed path"
                        nondeg += int(len(set(want.spl.tolist())) > 2)
    assert nondeg > 40
    # a weight with a pin term is NOT a width weight: still the oracle's answer (one-wave literal kernel)
    A = suitesparse_shaped(400, 5, 2)
    f = cp.ConstrainedCost(net, cp.AffineWorkModel(0, 1, 1), 700)
    assert cp.partition_stripe(A, 4, cp.DynamicTotalSplitter(f), backend=hip) == cp.partition_stripe(A, 4, cp.DynamicTotalSplitter(f), backend=orc)


def test_many_layers_on_one_handle(hip, orc):
    """K = 300 layers through one handle: the per-layer state that is NOT re-initialised every layer (the `fin` plane holds a layer
    stamp that wraps after 255 layers, csrc/dp_total.hip run_layer) must not leak from one layer into another"""
    A = suitesparse_shaped(2000, 8, 21)
    for mdl in (MODELS[1], MODELS[4]):
        for (K, w) in [(300, 600), (520, 130)]:
            f = cp.ConstrainedCost(mdl, cp.VertexCount(), w)
            got = cp.partition_stripe(A, K, cp.DynamicTotalSplitter(f), backend=hip)
            want = cp.partition_stripe(A, K, cp.DynamicTotalSplitter(f), backend=orc)
            assert got == want, (K, w)
    # the unconstrained layers too, with a gap-heavy previous row every time (tables compared cell by cell)
    mm = MODELS[1].marshal()
    rc1, p1, c1 = hip.dynamic_tables(A, 300, 0, mm, None)
    rc2, p2, c2 = orc.dynamic_tables(A, 300, 0, mm, None)
    assert rc1 == 0 and rc2 == 0 and np.array_equal(p1, p2) and np.array_equal(c1, c2)
