"""Pins the oracle's partitioners by re-expressing the reference's own property tests
(test_Partitioners.jl:76-302, test_Costs.jl:1-145) plus independent brute-force DPs written
from the recurrences on brute-force net tables."""
import numpy as np
import pytest

from util import (cp, sprand, dense_mask, net_table, selfnet_table, golden_matrices, suitesparse_shaped)

INF = float("inf")


def model_value(mdl, D, pos, T, S, j, jp, k):
    nv, npins = jp - j, int(pos[jp - 1] - pos[j - 1])
    if isinstance(mdl, cp.AffineWorkModel):
        return mdl(nv, npins, k)
    if isinstance(mdl, (cp.AffineConnectivityModel, cp.ColumnBlockComponentCostModel)):
        return mdl(nv, npins, int(T[j, jp]), k)
    if isinstance(mdl, cp.AffineHyperedgeCutModel):
        return mdl(nv, npins, int(S[j, jp]), int(T[j, jp] - S[j, jp]), k)
    raise TypeError


def brute_splitter(A, K, mdl, g):
    """cst[j',k] = min_j g(cst[j,k-1], f(j,j',k)); ties -> LARGEST j (DynamicSplitter.jl:36-43)."""
    D = dense_mask(A); T = net_table(D); S = selfnet_table(D); n = A.n
    f = lambda j, jp, k: model_value(mdl, D, A.colptr, T, S, j, jp, k)
    comb = (lambda a, b: a + b) if g == "sum" else max
    cst = {(jp, 1): f(1, jp, 1) for jp in range(1, n + 2)}
    ptr = {(jp, 1): 1 for jp in range(1, n + 2)}
    for k in range(2, K + 1):
        for jp in range(1, n + 2):
            best, arg = None, None
            for j in range(1, jp + 1):
                c = comb(cst[(j, k - 1)], f(j, jp, k))
                if best is None or c <= best:
                    best, arg = c, j
            cst[(jp, k)], ptr[(jp, k)] = best, arg
    spl = [0] * (K + 1)
    spl[K] = n + 1
    for k in range(K, 0, -1):
        spl[k - 1] = ptr[(spl[k], k)]
    return spl, cst[(n + 1, K)]


def brute_pack(A, mdl, wfun, w_max):
    """cst[j'] = min_{j0<=j<j'} cst[j] + f(j,j'); ties -> SMALLEST j (DynamicChunker.jl:41-49)."""
    D = dense_mask(A); T = net_table(D); S = selfnet_table(D); n = A.n
    f = lambda j, jp: model_value(mdl, D, A.colptr, T, S, j, jp, None)
    cst = {1: 0}; spl = {}
    for jp in range(2, n + 2):
        best, arg = None, None
        for j in range(1, jp):
            if wfun is not None and wfun(j, jp) > w_max:
                continue
            c = cst[j] + f(j, jp)
            if best is None or c < best:
                best, arg = c, j
        cst[jp], spl[jp] = best, arg
    out = [n + 1]
    while out[-1] != 1:
        out.append(spl[out[-1]])
    return out[::-1], cst[n + 1]


def small_matrices(seed, trials=2):
    rng = np.random.default_rng(seed)
    mats = [golden_matrices()["LPnetlib/lpi_itest6"]]
    for m in (1, 2, 3, 4, 8):
        for n in (1, 2, 3, 4, 8):
            for _ in range(trials):
                mats.append(sprand(m, n, 0.3, rng))
    mats.append(sprand(12, 14, 0.25, rng))
    return mats


MODELS = [
    cp.AffineWorkModel(0, 10, 1),
    cp.AffineConnectivityModel(0, 3, 1, 3),
    cp.AffineConnectivityModel(0, 0, 0, 1),
    cp.AffineConnectivityModel(0.0, 0.0, 0.0, 1.0),
    cp.AffineConnectivityModel(-0.5, 0.0, 0.0, 1.0),
    cp.AffineHyperedgeCutModel(0, 0, 0, 0, 1),
    cp.AffineHyperedgeCutModel(0, 1, 1, 1, 3),
]


@pytest.mark.parametrize("g", ["sum", "max"])
def test_dynamic_splitter_equals_bruteforce(orc, g):
    rng = np.random.default_rng(3)
    for A in small_matrices(10):
        for K in (1, 2, 3, 4, 8):
            funky = cp.AffineConnectivityModel(0, 3, 1, 3, alpha_k=rng.integers(1, 11, K).tolist())
            for mdl in MODELS + [funky]:
                meth = cp.DynamicTotalSplitter(mdl) if g == "sum" else cp.DynamicBottleneckSplitter(mdl)
                got = cp.partition_stripe(A, K, meth, backend=orc)
                want, val = brute_splitter(A, K, mdl, g)
                assert got.spl.tolist() == want, (A, K, mdl.kind, g)
                obj = (cp.total_value if g == "sum" else cp.bottleneck_value)(A, got, mdl, backend=orc)
                assert obj == pytest.approx(val, rel=1e-12)
                # chunker loop order (DynamicSplitter.jl:52-87) computes the same recurrence for
                # k-independent costs
                if mdl.alpha_k is None:
                    meth2 = cp.DynamicTotalChunker(mdl) if g == "sum" else cp.DynamicBottleneckChunker(mdl)
                    got2 = cp.partition_stripe(A, K, meth2, backend=orc)
                    assert got2.spl.tolist() == want


def test_structural_invariants_and_bisect(orc):
    """test_Partitioners.jl:95-113: sortedness, end points, K, bottleneck <= (1+eps) * Reference."""
    mats = small_matrices(20, trials=1) + [golden_matrices()["HB/can_292"]]
    for A in mats:
        for K in (1, 2, 3, 4, 8):
            for f in (cp.AffineWorkModel(0, 10, 1), cp.AffineConnectivityModel(0, 3, 1, 3)):
                if A.n > 100 and K > 4:
                    continue
                ref = cp.partition_stripe(A, K, cp.ReferenceBottleneckSplitter(f), backend=orc)
                c = cp.bottleneck_value(A, ref, f, backend=orc)
                for eps in (0.1, 0.01):
                    got = cp.partition_stripe(A, K, cp.BisectCostBottleneckSplitter(f, eps), backend=orc)
                    s = got.spl
                    assert np.all(np.diff(s) >= 0) and s[0] == 1 and s[-1] == A.n + 1 and got.K == K
                    assert cp.bottleneck_value(A, got, f, backend=orc) <= c * (1 + eps)
                lo, hi = cp.bound_stripe(A, K, f, backend=orc)
                assert 0 <= lo <= c <= hi                       # test_Costs.jl:29-30


def test_flip_bisect(orc):
    """Monotone-decreasing costs (test_Partitioners.jl:116-152): per-part alpha with negative betas."""
    rng = np.random.default_rng(4)
    for A in small_matrices(21, trials=1):
        for K in (1, 2, 3, 4):
            base = 1 + A.nnz + 3 * A.n + 3 * A.m
            f = cp.AffineConnectivityModel(0, -3, -1, -3, alpha_k=(base + rng.integers(1, 11, K)).tolist())
            ref = cp.partition_stripe(A, K, cp.ReferenceBottleneckSplitter(f), backend=orc)
            c = cp.bottleneck_value(A, ref, f, backend=orc)
            got, _ = brute_splitter(A, K, f, "max")
            assert ref.spl.tolist() == got


def test_pack_dynamic_equals_bruteforce(orc):
    for A in small_matrices(30):
        if A.n < 1:
            continue
        for f in (cp.AffineConnectivityModel(0, 3, 1, 3), cp.AffineConnectivityModel(-0.5, 0.0, 0.0, 1.0),
                  cp.AffineWorkModel(0, 0, 0)):
            got = cp.pack_stripe(A, cp.DynamicTotalChunker(f), backend=orc)
            want, val = brute_pack(A, f, None, None)
            assert got.spl.tolist() == want
            assert cp.total_value(A, got, f, backend=orc) == pytest.approx(val)
            for w_max in (2, 4, 8):
                fc = cp.ConstrainedCost(f, cp.VertexCount(), w_max)
                got = cp.pack_stripe(A, cp.DynamicTotalChunker(fc), backend=orc)
                want, val = brute_pack(A, f, lambda j, jp: jp - j, w_max)
                assert got.spl.tolist() == want
                assert np.all(np.diff(got.spl) <= w_max)                     # test_Partitioners.jl:243
                fw = cp.ConstrainedCost(f, cp.AffineWorkModel(0, 1, 0), w_max)
                got2 = cp.pack_stripe(A, cp.DynamicTotalChunker(fw), backend=orc)
                assert got2.spl.tolist() == want


def test_convex_matches_dp_optimum(orc):
    """test_Partitioners.jl:174-198, 250-275: Convex{Splitter,Chunker} reach the DP's total value."""
    mats = small_matrices(40, trials=1) + [golden_matrices()["LPnetlib/lp_blend"], suitesparse_shaped(60, 4, 1)]
    fs = [cp.AffineConnectivityModel(0.0, 0.0, 0.0, 1.0), cp.AffineConnectivityModel(0, 0, 0, 1),
          cp.AffineConnectivityModel(-0.5, 0.0, 0.0, 1.0), cp.AffineWorkModel(0, 0, 0),
          cp.AffineConnectivityModel(0, 3, 1, 3)]
    # NB ColumnBlockComponentCostModel(3, w->1+w) is NOT inverse-Monge (nets * width), so the
    # stack algorithm is not guaranteed optimal for it; its output is pinned by GPU-vs-oracle parity only.
    for A in mats:
        for f in fs:
            ref = cp.pack_stripe(A, cp.ReferenceTotalChunker(f), backend=orc)
            c = cp.total_value(A, ref, f, backend=orc)
            got = cp.pack_stripe(A, cp.ConvexTotalChunker(f), backend=orc)
            assert got.spl[0] == 1 and got.spl[-1] == A.n + 1 and np.all(np.diff(got.spl) >= 0)
            assert cp.total_value(A, got, f, backend=orc) == pytest.approx(c)
            for w_max in (2, 4, 8):
                fc = cp.ConstrainedCost(f, cp.AffineWorkModel(0, 1, 0), w_max)
                ref = cp.pack_stripe(A, cp.ReferenceTotalChunker(fc), backend=orc)
                c2 = cp.total_value(A, ref, f, backend=orc)
                got = cp.pack_stripe(A, cp.ConvexTotalChunker(fc), backend=orc)
                assert np.all(np.diff(got.spl) <= w_max) and got.spl[-1] == A.n + 1
                assert cp.total_value(A, got, f, backend=orc) == pytest.approx(c2)
            for K in (1, 2, 3, 4, 8):
                if A.n > 60 and K > 3:
                    continue
                ref = cp.partition_stripe(A, K, cp.ReferenceTotalSplitter(f), backend=orc)
                c = cp.total_value(A, ref, f, backend=orc)
                got = cp.partition_stripe(A, K, cp.ConvexTotalSplitter(f), backend=orc)
                assert got.spl[0] == 1 and got.spl[-1] == A.n + 1 and np.all(np.diff(got.spl) >= 0)
                assert cp.total_value(A, got, f, backend=orc) == pytest.approx(c)
                for w_max in (2, 4, 8):
                    fc = cp.ConstrainedCost(f, cp.AffineWorkModel(0, 1, 0), w_max)
                    ref = cp.partition_stripe(A, K, cp.ReferenceTotalSplitter(fc), backend=orc)
                    got = cp.partition_stripe(A, K, cp.ConvexTotalSplitter(fc), backend=orc)
                    got2 = cp.partition_stripe(A, K, cp.DynamicTotalChunker(fc), backend=orc)
                    if np.all(np.diff(ref.spl) <= w_max):          # feasible
                        cr = cp.total_value(A, ref, f, backend=orc)
                        assert cp.total_value(A, got, f, backend=orc) == pytest.approx(cr)
                        assert cp.total_value(A, got2, f, backend=orc) == pytest.approx(cr)
                    else:                                           # degenerate [1,..,1,n+1]
                        assert ref.spl.tolist() == [1] * K + [A.n + 1]
                        assert got.spl.tolist() == ref.spl.tolist()


def block_total_direct(A, Pi, Phi, mdl):
    """Direct evaluation of the rank-R block cost: per column part, sum over row parts touching it."""
    D = dense_mask(A)
    bc = lambda f, w: f(w) if callable(f) else f
    tot = 0
    asg = cp.to_map(Pi).asg
    for k in range(Phi.K):
        j, jp = Phi.spl[k], Phi.spl[k + 1]
        w = jp - j
        c = bc(mdl.alpha_col, w)
        touched = np.unique(asg[D[:, j - 1:jp - 1].any(axis=1)])
        for r in range(len(mdl.beta_row)):
            d = sum(bc(mdl.beta_row[r], int(Pi.spl[kk] - Pi.spl[kk - 1])) for kk in touched)
            c += d * bc(mdl.beta_col[r], w)
        tot += c
    return tot


def test_block_costs(orc):
    """test_Costs.jl:106-122 analogue: BlockComponentCostStepOracle total == direct evaluation."""
    rng = np.random.default_rng(6)
    for m in (3, 8, 17, 30):
        for u in (1, 2, 3, 4):
            for w in (1, 2, 3, 4):
                A = sprand(m, m, 0.125, rng)
                Pi = cp.pack_stripe(A, cp.EquiChunker(u))
                Phi = cp.pack_stripe(A, cp.EquiChunker(w))
                for mdl in (cp.BlockComponentCostModel(0, 0, (2, lambda x: x), (2, lambda x: 2 * x)),
                            cp.BlockComponentCostModel(lambda x: x, lambda x: 3 * x, (2, lambda x: x), (2, lambda x: 2 * x))):
                    got = cp.total_value(A, Phi, mdl, Pi, backend=orc)
                    assert got == block_total_direct(A, Pi, Phi, mdl)
    # chunkers with a block model: DP optimum respected, width limit honoured (test_Partitioners.jl:225-248)
    A = sprand(20, 24, 0.2, rng)
    Pi = cp.pack_stripe(cp.adjointpattern(A, backend=orc), cp.EquiChunker(2))
    mdl = cp.BlockComponentCostModel(0, 0, (10, lambda x: x), (2, lambda x: 2 * x))
    f = cp.ConstrainedCost(mdl, cp.VertexCount(), 4)
    Phi = cp.pack_stripe(A, cp.DynamicTotalChunker(f), Pi, backend=orc)
    assert np.all(np.diff(Phi.spl) <= 4) and Phi.spl[-1] == A.n + 1
    best = cp.total_value(A, Phi, mdl, Pi, backend=orc)
    for w in (1, 2, 3, 4):
        other = cp.pack_stripe(A, cp.EquiChunker(w))
        assert cp.total_value(A, other, mdl, Pi, backend=orc) >= best


def test_equi(cp_=None):
    A = sprand(5, 17, 0.3, np.random.default_rng(0))
    assert cp.partition_stripe(A, 4, cp.EquiSplitter()).spl.tolist() == [1, 6, 10, 14, 18]
    assert cp.pack_stripe(A, cp.EquiChunker(5)).spl.tolist() == [1, 6, 11, 16, 18]
