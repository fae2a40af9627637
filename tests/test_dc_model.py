"""The device algorithm's executable specification (tests/dc_model.py) reproduces the literal
DP tables of the oracle: every cst[j', k] and ptr[j', k], ties included."""
import numpy as np
import pytest

from util import cp, sprand, golden_matrices, suitesparse_shaped
import dc_model


def check(A, K, mdl, orc):
    mm = mdl.marshal()
    rc, ptr, cst = orc.dynamic_tables(A, K, 0, mm, None)
    assert rc == 0
    prev, nxt = dc_model.link_arrays(A)
    pos = A.colptr - 1
    n = A.n
    for k in range(2, K):                     # layers 2..K-1 are complete in the reference tables
        alpha = mdl.alpha if mdl.alpha_k is None else mdl.alpha_k[k - 1]

        def fcost(p, r, nn, alpha=alpha):
            return alpha + (r - p) * mdl.beta_vertex + int(pos[r] - pos[p]) * mdl.beta_pin + nn * mdl.beta_net
        W = [cst[j, k - 2].item() for j in range(n + 1)]
        c2, p2 = dc_model.layer_total(A, W, fcost, prev, nxt)
        assert [int(x) for x in p2 + 1] == ptr[:, k - 1].tolist(), (A, k)
        assert [float(x) for x in c2] == [float(x) for x in cst[:, k - 1]]


def test_dc_scheme_reproduces_literal_tables(orc):
    rng = np.random.default_rng(123)
    mats = [sprand(m, n, p, rng) for (m, n, p) in
            [(5, 7, 0.4), (8, 16, 0.3), (10, 23, 0.2), (6, 33, 0.3), (20, 40, 0.1), (3, 12, 0.6), (12, 31, 0.15)]]
    mats += [golden_matrices()["LPnetlib/lpi_itest6"], suitesparse_shaped(50, 3, 5)]
    for A in mats:
        for mdl in (cp.AffineConnectivityModel(0, 0, 0, 1), cp.AffineConnectivityModel(0, 10, 1, 100),
                    cp.AffineConnectivityModel(0, 3, 1, 3), cp.AffineConnectivityModel(2, 0, 1, 1),
                    cp.AffineConnectivityModel(0, 3, 1, 3, alpha_k=[5, 1, 9, 2, 7])):
            check(A, 5, mdl, orc)


def test_dc_scheme_hyperedge_cut(orc):
    """Hyperedge-cut costs d*b_cut + l*(b_self - b_cut) are inverse-Monge when b_cut >= 0 and b_self <= b_cut
    (SURVEY.md section 7): the scheme reproduces the literal tables with the (nets, selfnets) pair of counts."""
    rng = np.random.default_rng(321)
    mats = [sprand(m, n, p, rng) for (m, n, p) in [(5, 7, 0.4), (8, 16, 0.3), (10, 23, 0.2), (6, 33, 0.3), (20, 40, 0.1), (12, 31, 0.15)]]
    mats += [golden_matrices()["LPnetlib/lpi_itest6"], suitesparse_shaped(40, 3, 9)]
    K = 5
    for A in mats:
        prev, nxt = dc_model.link_arrays(A)
        first, last = dc_model.first_last(A)
        pos = A.colptr - 1
        n = A.n
        for (bs, bc) in [(0, 1), (0, 3), (1, 1), (-1, 0), (1, 3), (2, 5)]:
            mdl = cp.AffineHyperedgeCutModel(0, 2, 1, bs, bc)
            rc, ptr, cst = orc.dynamic_tables(A, K, 0, mdl.marshal(), None)
            assert rc == 0
            for k in range(2, K):
                def fcost(p, r, c):
                    nn, nl = int(c[0]), int(c[1])
                    return mdl(r - p, int(pos[r] - pos[p]), nl, nn - nl)
                W = [cst[j, k - 2].item() for j in range(n + 1)]
                c2, p2 = dc_model.layer_total(A, W, fcost, prev, nxt, first, last)
                assert [int(x) for x in p2 + 1] == ptr[:, k - 1].tolist(), (A, k, bs, bc)
                assert [int(x) for x in c2] == [int(x) for x in cst[:, k - 1]]


def _w_rows(rng, n, scale):
    """previous-layer rows that are NOT the closed-form row: arbitrary, monotone, and tie-heavy ones"""
    return [rng.integers(0, scale + 1, n + 1), np.sort(rng.integers(0, scale + 1, n + 1)),
            np.sort(rng.integers(0, scale + 1, n + 1))[::-1].copy(), rng.integers(0, 3, n + 1),
            np.zeros(n + 1, dtype=np.int64)]


def test_brute_force_layer_is_the_oracles_layer(orc):
    """tests/brute.py (counts from their definitions, the recurrence as written) reproduces the oracle's tables layer by
    layer -- it may then stand in for the oracle where a previous-layer row is injected."""
    import brute
    rng = np.random.default_rng(7)
    for A in [sprand(8, 16, 0.3, rng), sprand(10, 23, 0.2, rng), golden_matrices()["LPnetlib/lpi_itest6"], suitesparse_shaped(60, 3, 5)]:
        for mdl in (cp.AffineConnectivityModel(0, 3, 1, 3), cp.AffineHyperedgeCutModel(0, 2, 1, 1, 3), cp.AffineWorkModel(1, 10, 1),
                    cp.AffineConnectivityModel(0, 3, 1, 3, alpha_k=[5, 1, 9, 2])):
            K = 4
            rc, ptr, cst = orc.dynamic_tables(A, K, 0, mdl.marshal(), None)
            assert rc == 0
            for k in range(2, K):
                F = brute.cost_table(A, mdl, k)
                c2, p2 = brute.layer(cst[:, k - 2], F)
                assert np.array_equal(c2, cst[:, k - 1]) and np.array_equal(p2 + 1, ptr[:, k - 1])


def test_dc_scheme_block_argmins_with_arbitrary_previous_rows():
    """The closed form ptr[j'] = j' hides the per-block winners of the scheme for every true previous layer (the diagonal
    candidate always ties the minimum).  W[p] + f(p, r) stays inverse-Monge for ANY row W, so inject rows that move the
    arg-mins off the diagonal and compare BOTH the combined result and every per-block winner with brute force."""
    import brute
    rng = np.random.default_rng(99)
    mats = [sprand(5, 7, 0.4, rng), sprand(8, 16, 0.3, rng), sprand(10, 23, 0.2, rng), sprand(6, 33, 0.3, rng), sprand(20, 40, 0.1, rng),
            sprand(12, 31, 0.15, rng), suitesparse_shaped(50, 3, 5)]
    nontrivial = 0
    for A in mats:
        n = A.n
        prev, nxt = dc_model.link_arrays(A)
        first, last = dc_model.first_last(A)
        pos = A.colptr - 1
        nb = max(1, int(n).bit_length())
        for mdl in (cp.AffineConnectivityModel(0, 0, 0, 1), cp.AffineConnectivityModel(1, 2, 1, 3), cp.AffineHyperedgeCutModel(0, 2, 1, 1, 3)):
            hyper = mdl.kind == cp.models.CP_MODEL_HYPEREDGE_CUT
            F = brute.cost_table(A, mdl)
            if hyper:
                def fcost(p, r, c):
                    nn, nl = int(c[0]), int(c[1])
                    return mdl(r - p, int(pos[r] - pos[p]), nl, nn - nl)
            else:
                def fcost(p, r, nn):
                    return mdl(r - p, int(pos[r] - pos[p]), nn)
            for W in _w_rows(rng, n, int(F.max()) + 1):
                Wl = [int(x) for x in W]
                c2, p2, o2 = dc_model.layer_total(A, Wl, fcost, prev, nxt, *((first, last) if hyper else ()), blocks=True)
                cb, pb = brute.layer(W, F)
                assert [int(x) for x in c2] == cb.tolist() and p2.tolist() == pb.tolist()
                ob = brute.block_argmins(W, F, nb)
                assert np.array_equal(o2[:nb], ob), (A, mdl.kind)
                nontrivial += int(np.sum(pb != np.arange(n + 1)))
    assert nontrivial > 300          # the injected rows do move the winners off the diagonal


def test_windowed_scheme_equals_brute_force():
    """The width-windowed geometry (standard / common / mirrored planes) reproduces min over max(0, r-w) <= p <= r with the
    largest minimiser, for arbitrary previous rows -- including rows masked with a huge value outside a window, as the
    constrained DP's previous layer is."""
    import brute
    rng = np.random.default_rng(2024)
    mats = [sprand(5, 7, 0.4, rng), sprand(8, 16, 0.3, rng), sprand(10, 23, 0.2, rng), sprand(6, 33, 0.3, rng), sprand(20, 40, 0.1, rng),
            sprand(12, 31, 0.15, rng), suitesparse_shaped(50, 3, 5), sprand(9, 64, 0.2, rng), sprand(9, 65, 0.2, rng)]
    BIG = 1 << 40
    moved = 0
    for A in mats:
        n = A.n
        prev, nxt = dc_model.link_arrays(A)
        first, last = dc_model.first_last(A)
        pos = A.colptr - 1
        for mdl in (cp.AffineConnectivityModel(0, 0, 0, 1), cp.AffineConnectivityModel(1, 2, 1, 3), cp.AffineHyperedgeCutModel(0, 2, 1, 1, 3)):
            hyper = mdl.kind == cp.models.CP_MODEL_HYPEREDGE_CUT
            F = brute.cost_table(A, mdl)
            if hyper:
                def fcost(p, r, c):
                    nn, nl = int(c[0]), int(c[1])
                    return mdl(r - p, int(pos[r] - pos[p]), nl, nn - nl)
            else:
                def fcost(p, r, nn):
                    return mdl(r - p, int(pos[r] - pos[p]), nn)
            for w in sorted(set([1, 2, 3, 4, 5, 7, 8, 11, max(1, n // 3), max(1, n // 2), n - 1, n, n + 5])):
                if w < 1:
                    continue
                rows = _w_rows(rng, n, int(F.max()) + 1)[:3]
                masked = rows[0].copy()
                a, b = sorted(rng.integers(0, n + 1, 2))
                masked[:a] = BIG; masked[b + 1:] = BIG
                for W in rows + [masked]:
                    Wl = [int(x) for x in W]
                    c2, p2 = dc_model.layer_windowed(A, Wl, fcost, prev, nxt, w, *((first, last) if hyper else ()))
                    lo = np.maximum(0, np.arange(n + 1) - w)
                    cb, pb = brute.layer(W, F, lo=lo)
                    assert p2.tolist() == pb.tolist(), (A, w, mdl.kind)
                    assert [int(x) for x in c2] == cb.tolist()
                    moved += int(np.sum(pb != np.arange(n + 1)))
    assert moved > 1000


def test_windowed_scheme_composes_to_the_constrained_splitter(orc):
    """Layer by layer, the windowed scheme fed with the previous layer masked outside its window [j'_lo, j'_hi]
    (column_constraints, DynamicSplitter.jl:144-172) reproduces the oracle's ConstrainedCost splitter tables
    (DynamicSplitter.jl:206-258) -- every in-window cst / ptr cell, and non-degenerate split vectors."""
    rng = np.random.default_rng(31)
    BIG = 1 << 50
    mats = [sprand(8, 16, 0.3, rng), sprand(10, 23, 0.2, rng), sprand(6, 33, 0.3, rng), sprand(20, 40, 0.1, rng), suitesparse_shaped(60, 3, 5),
            golden_matrices()["LPnetlib/lpi_itest6"]]
    nondegenerate = 0
    for A in mats:
        n = A.n
        prev, nxt = dc_model.link_arrays(A)
        pos = A.colptr - 1
        for mdl in (cp.AffineConnectivityModel(0, 0, 0, 1), cp.AffineConnectivityModel(0, 3, 1, 3, alpha_k=[5, 1, 9, 2, 7])):
            for K in (2, 3, 5):
                for w in sorted(set([max(1, -(-n // K)), max(1, -(-3 * n // (2 * K))), max(1, n // 2), n])):
                    wm = cp.VertexCount().marshal()
                    rc, lo, hi, ptr, cst = orc.dynamic_tables_constrained(A, K, 0, mdl.marshal(), None, wm, w, float(w))
                    if rc == 2:
                        assert hi[K - 1] < n + 1
                        continue
                    assert rc == 0
                    # windows: the closed form of column_constraints for the width weight
                    assert lo.tolist() == [max(1, n + 1 - (K - k) * w) for k in range(1, K + 1)]
                    assert hi.tolist() == [min(n + 1, 1 + k * w) for k in range(1, K + 1)]
                    Wk = None
                    for k in range(1, K + 1):
                        a, b = lo[k - 1] - 1, hi[k - 1] - 1
                        alpha = mdl.alpha if mdl.alpha_k is None else mdl.alpha_k[k - 1]

                        def fcost(p, r, nn, alpha=alpha):
                            return alpha + (r - p) * mdl.beta_vertex + int(pos[r] - pos[p]) * mdl.beta_pin + nn * mdl.beta_net
                        if k == 1:
                            NT0 = [len(set(A.rowval[:pos[r]].tolist())) for r in range(n + 1)]
                            c2 = [fcost(0, r, NT0[r]) for r in range(n + 1)]
                            p2 = np.zeros(n + 1, dtype=np.int64)
                        else:
                            c2, p2 = dc_model.layer_windowed(A, Wk, fcost, prev, nxt, w)
                        assert [int(x) for x in c2[a:b + 1]] == cst[a:b + 1, k - 1].tolist(), (A, K, w, k)
                        assert (p2[a:b + 1] + 1).tolist() == ptr[a:b + 1, k - 1].tolist(), (A, K, w, k)
                        assert not ptr[:a, k - 1].any() and not ptr[b + 1:, k - 1].any()
                        Wk = [int(c2[p]) if a <= p <= b else BIG for p in range(n + 1)]
                    spl = cp.partition_stripe(A, K, cp.DynamicTotalSplitter(cp.ConstrainedCost(mdl, cp.VertexCount(), w)), backend=orc).spl
                    nondegenerate += int(len(set(spl.tolist())) > 2)
    assert nondegenerate > 20
