"""Checks of the one-wave sequential kernels (csrc/seq.hip) that share NOTHING with their text: seq.hip follows the reference's
stack / deque algorithms statement for statement, and so does the oracle, so a transliteration slip would be common to both.
Here the GPU results are held against the problem's DEFINITION instead (tests/brute.py: counts and costs from their definitions,
the optimum by exhaustive DP) and the structural assertions of the reference's own tests (test_Partitioners.jl:185-299):
  issorted(spl), spl[1] == 1, spl[end] == n + 1, widths <= w_max, total_value == the optimum for convex (inverse-Monge) costs on
  the Convex methods and concave (Monge) costs on the Concave methods."""
import numpy as np
import pytest

import brute
from util import cp, sprand, golden_matrices, suitesparse_shaped

pytestmark = pytest.mark.gpu


def mats():
    rng = np.random.default_rng(0xDEADBEEF)
    out = [sprand(m, n, 0.1, rng) for (m, n) in [(1, 1), (2, 3), (3, 2), (4, 8), (8, 4), (8, 8)]]      # test_Partitioners.jl:77-83
    out += [golden_matrices()["LPnetlib/lpi_itest6"], golden_matrices()["HB/can_292"], suitesparse_shaped(300, 4, 2)]
    return out


def structure_ok(spl, n, wmax=None):
    return spl[0] == 1 and spl[-1] == n + 1 and np.all(np.diff(spl) >= 0) and (wmax is None or np.all(np.diff(spl) <= wmax))


def close(a, b):
    return a == b or abs(a - b) <= 1e-12 * max(abs(a), abs(b))        # Float64 totals within 1e-12 relative


CONVEX = [(cp.ConvexWorkModel(0.0, 0, 1), None), (cp.ConvexWorkModel(-0.7, 0, 1), None), (cp.AffineConnectivityModel(-0.5, 0.0, 0.0, 1.0), None),
          (cp.AffineConnectivityModel(0, 0, 0, 1), None), (cp.AffineWorkModel(0, 0, 0), None), (cp.AffineConnectivityModel(0, 0, 0, 1), 2),
          (cp.AffineConnectivityModel(0, 0, 0, 1), 4), (cp.AffineConnectivityModel(0, 0, 0, 1), 8), (cp.ConvexWorkModel(0, 1, 0), 2),
          (cp.ConvexWorkModel(0, 1, 0), 4), (cp.ConvexWorkModel(0, 0, 1), 8)]                               # test_Partitioners.jl:251-262
CONCAVE = [(cp.ConcaveWorkModel(0.0, 0, 1), None), (cp.ConcaveWorkModel(-0.7, 0, 1), None), (cp.AffineWorkModel(0, 0, 0), None),
           (cp.ConcaveWorkModel(0, 1, 0), 2), (cp.ConcaveWorkModel(0, 1, 0), 4), (cp.ConcaveWorkModel(0, 0, 1), 8)]   # :278-284


def costed(mdl, wmax):
    return mdl if wmax is None else cp.ConstrainedCost(mdl, cp.AffineWorkModel(0, 1, 0), wmax)          # width via the work model, as the reference's tests do


@pytest.mark.parametrize("family", ["convex", "concave"])
def test_chunkers_reach_the_brute_force_optimum(hip, family):
    Meth = cp.ConvexTotalChunker if family == "convex" else cp.ConcaveTotalChunker
    for A in mats():
        for mdl, wmax in (CONVEX if family == "convex" else CONCAVE):
            F = brute.cost_table(A, mdl)
            opt = brute.lws_optimum(F.astype(np.float64), wmax)
            for M_ in (Meth, cp.DynamicTotalChunker):
                P = cp.pack_stripe(A, M_(costed(mdl, wmax)), backend=hip)
                assert structure_ok(P.spl, A.n, wmax), (A, family, M_.__name__)
                v = brute.partition_value(lambda k: F, P.spl)
                assert close(float(v), float(opt)), (A, family, M_.__name__, v, opt)


@pytest.mark.parametrize("family", ["convex", "concave"])
def test_splitters_reach_the_brute_force_optimum(hip, family):
    Meth = cp.ConvexTotalSplitter if family == "convex" else cp.ConcaveTotalSplitter
    models = (CONVEX if family == "convex" else CONCAVE)
    for A in mats():
        for K in (1, 2, 3, 4):
            for mdl, wmax in models:
                if wmax is not None and K * wmax < A.n:
                    continue                                   # infeasible windows: the degenerate partition, nothing to optimise
                F = brute.cost_table(A, mdl).astype(np.float64)
                opt = brute.kpart_optimum(lambda k: F, K, wmax)
                for M_ in (Meth, cp.DynamicTotalSplitter):
                    P = cp.partition_stripe(A, K, M_(costed(mdl, wmax)), backend=hip)
                    assert structure_ok(P.spl, A.n, wmax) and len(P.spl) == K + 1, (A, K, family, M_.__name__)
                    v = brute.partition_value(lambda k: F, P.spl)
                    assert close(float(v), float(opt)), (A, K, family, M_.__name__, v, opt)


def test_block_cost_oracle_equals_its_definition(hip):
    """BlockComponentCostModel (BlockCosts.jl:19-152): cost(j, j') = alpha_col(w) + sum_r d[r] * beta_col[r](w), w = j' - j,
    d[r] = sum over the row parts k holding a nonzero of columns j .. j'-1 of beta_row[r](|part k|) -- evaluated here from the
    dense pattern, for random query sequences (the device oracle is stateful: order matters to its internals, not to its values)"""
    rng = np.random.default_rng(3)
    from util import dense_mask
    for A in [sprand(8, 16, 0.3, rng), sprand(20, 12, 0.2, rng), golden_matrices()["LPnetlib/lpi_itest6"]]:
        D = dense_mask(A)
        Pi = cp.pack_stripe(cp.adjointpattern(A), cp.EquiChunker(2))
        for mdl in (cp.BlockComponentCostModel(0, 0, (10, lambda u: u), (2, lambda x: 2 * x)),
                    cp.BlockComponentCostModel(lambda u: u, lambda x: 3 * x, (10, lambda u: u), (2, lambda x: 2 * x))):
            ocl = cp.oracle_stripe(cp.StepHint(), mdl, A, Pi, backend=hip)
            js = rng.integers(1, A.n + 2, 200); jps = np.array([rng.integers(j, A.n + 2) for j in js])
            got = ocl(js, jps)
            fn = lambda f, x: f(x) if callable(f) else f
            for t in range(len(js)):
                j, jp = int(js[t]), int(jps[t])
                w = jp - j
                touched = D[:, j - 1:jp - 1].any(axis=1)
                c = fn(mdl.alpha_col, w)
                for r in range(len(mdl.beta_row)):
                    d = 0
                    for k in range(Pi.K):
                        rows = slice(Pi.spl[k] - 1, Pi.spl[k + 1] - 1)
                        if touched[rows].any():
                            d += fn(mdl.beta_row[r], Pi.spl[k + 1] - Pi.spl[k])
                    c += d * fn(mdl.beta_col[r], w)
                assert got[t] == c, (A, j, jp)
