"""Executable statement of the valley search behind csrc/dp_bottleneck.hip, checked against the oracle's literal bottleneck DP
(DynamicSplitter.jl:7,33-46 with g = max): for costs that grow with their part, cst[r] = min over p <= r of max(W[p], f(p, r)) is
min(f(c-1, r), W[c]) at the crossing c = min{p : W[p] >= f(p, r)}, and the reference's "largest j on ties" is the right end of
W's run through c (capped at r) when W[c] <= f(c-1, r), else c - 1."""
import numpy as np

import brute
from util import cp, sprand, golden_matrices, suitesparse_shaped


def valley_layer(W, F):
    n1 = F.shape[0]
    cst = np.zeros(n1, dtype=F.dtype); ptr = np.zeros(n1, dtype=np.int64)
    runend = np.zeros(n1, dtype=np.int64)
    e = n1 - 1
    for p in range(n1 - 1, -1, -1):
        if p < n1 - 1 and W[p] != W[p + 1]:
            e = p
        runend[p] = e
    c = 0
    for r in range(n1):
        while c <= r and W[c] < F[c, r]:       # the crossing only moves right
            c += 1
        fm = F[c - 1, r] if c >= 1 else None
        if c <= r and (fm is None or W[c] <= fm):
            cst[r], ptr[r] = W[c], min(runend[c], r)
        else:
            cst[r], ptr[r] = fm, c - 1
    return cst, ptr


def test_valley_search_reproduces_the_literal_bottleneck_tables(orc):
    rng = np.random.default_rng(5)
    mats = [sprand(8, 16, 0.3, rng), sprand(10, 23, 0.2, rng), sprand(6, 33, 0.3, rng), sprand(20, 40, 0.1, rng), suitesparse_shaped(60, 3, 5),
            sprand(3, 12, 0.6, rng), golden_matrices()["LPnetlib/lpi_itest6"]]
    K = 5
    for A in mats:
        for mdl in (cp.AffineConnectivityModel(0, 0, 0, 1), cp.AffineConnectivityModel(2, 3, 1, 3), cp.AffineWorkModel(1, 1, 0),
                    cp.AffineHyperedgeCutModel(0, 1, 0, 3, 2), cp.AffineHyperedgeCutModel(0, 0, 0, 1, 1),
                    cp.AffineConnectivityModel(0, 3, 1, 3, alpha_k=[5, 1, 9, 2, 7]), cp.AffineConnectivityModel(0.5, 0.25, 0.0, 1.5)):
            rc, ptr, cst = orc.dynamic_tables(A, K, 1, mdl.marshal(), None)
            assert rc == 0
            for k in range(2, K):                          # layers 2..K-1 are complete in the reference tables
                F = brute.cost_table(A, mdl, k)
                c1, p1 = valley_layer(cst[:, k - 2], F)
                assert np.array_equal(c1, cst[:, k - 1]) and np.array_equal(p1 + 1, ptr[:, k - 1]), (A, mdl.kind, k)


def _literal_max_layer(W, F):
    """DynamicSplitter.jl:33-46 with g = max as written: scan p upwards, `<=` keeps the largest minimiser"""
    n1 = F.shape[0]
    cst = np.zeros(n1, dtype=F.dtype); ptr = np.zeros(n1, dtype=np.int64)
    for r in range(n1):
        v = np.maximum(W[:r + 1], F[:r + 1, r])
        i = v.size - 1 - int(np.argmin(v[::-1]))
        cst[r], ptr[r] = v[i], i
    return cst, ptr


def test_non_integral_hyperedge_costs_are_outside_the_valley_class(orc):
    """Why fast_bottleneck_ok (csrc/capi.hip) sends Float64 hyperedge-cut models with non-integral betas to the general sweep
    (ADVICE round 2): the cost fl(l*b_self) + fl((d-l)*b_cut) is not monotone in the part after rounding, so the valley search
    is NOT the literal DP for them -- while a literal sweep over the same rounded cost table is.  Work / Connectivity models with
    non-integral parameters stay inside the class (every term is monotone)."""
    rng = np.random.default_rng(5)
    mats = [sprand(8, 16, 0.3, rng), sprand(10, 23, 0.2, rng), sprand(6, 33, 0.3, rng), sprand(20, 40, 0.1, rng), suitesparse_shaped(60, 3, 5),
            sprand(3, 12, 0.6, rng)]
    K = 5
    bad_valley = 0
    for A in mats:
        for mdl in (cp.AffineHyperedgeCutModel(0., 0., 0., 0.1, 0.1), cp.AffineHyperedgeCutModel(0., 0., 0., 0.7, 0.1),
                    cp.AffineHyperedgeCutModel(0.3, 0.1, 0., 0.3, 0.3)):
            rc, ptr, cst = orc.dynamic_tables(A, K, 1, mdl.marshal(), None)
            assert rc == 0
            for k in range(2, K):
                F = brute.cost_table(A, mdl, k)
                c0, p0 = _literal_max_layer(cst[:, k - 2], F)
                assert np.array_equal(c0, cst[:, k - 1]) and np.array_equal(p0 + 1, ptr[:, k - 1]), (A, k)      # the literal sweep IS the oracle
                c1, p1 = valley_layer(cst[:, k - 2], F)
                bad_valley += int(not (np.array_equal(c1, cst[:, k - 1]) and np.array_equal(p1 + 1, ptr[:, k - 1])))
        for mdl in (cp.AffineConnectivityModel(0.1, 0.7, 0.3, 0.1), cp.AffineWorkModel(0.3, 0.1, 0.7)):
            rc, ptr, cst = orc.dynamic_tables(A, K, 1, mdl.marshal(), None)
            for k in range(2, K):
                c1, p1 = valley_layer(cst[:, k - 2], brute.cost_table(A, mdl, k))
                assert np.array_equal(c1, cst[:, k - 1]) and np.array_equal(p1 + 1, ptr[:, k - 1]), (A, mdl.kind, k)
    assert bad_valley > 0          # (the advisor counted 35 of 360 layers on a similar set)
