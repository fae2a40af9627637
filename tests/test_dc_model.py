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
