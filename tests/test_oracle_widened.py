"""SURVEY 8(f) rows 1-2 on the CPU oracle: BisectIndexBottleneckSplitter (+Flip) and the connectivity
specialisation of LazyBisectCostBottleneckSplitter.

The C restatements are pinned two ways: (i) the reference's own property tests
(test_Partitioners.jl:95-152: sortedness, end points, K, bottleneck value against
ReferenceBottleneckSplitter with the method's epsilon) and (ii) a second, independent statement of the two
algorithms in Python on brute-force net tables, whose split vectors must match exactly."""
import numpy as np

from util import cp, dense_mask, net_table, selfnet_table, golden_matrices
from test_oracle_partitioners import model_value, small_matrices


def fld2(x):
    return x >> 1


def py_bisect_index(A, K, f, fn, bounds, flip):
    """BisectIndexBottleneckSplitter.jl:5-83 / :85-166, transcribed independently of oracle/orc_algos.inc."""
    n = A.n

    def search(j, lo, hi, k, c):
        lo = max(j, lo)
        while lo <= hi:
            jp = fld2(lo + hi)
            if fn(j, jp, k) <= c:
                lo = jp + 1
            else:
                hi = jp - 1
        return hi

    def search_flip(j, lo, hi, k, c):
        lo = max(j, lo)
        while lo <= hi:
            jp = fld2(lo + hi)
            if fn(j, jp, k) <= c:
                hi = jp - 1
            else:
                lo = jp + 1
        return lo

    spl_lo = [1] * (K + 1); spl_lo[K] = n + 1
    spl_hi = [n + 1] * (K + 1); spl_hi[0] = 1
    spl = [0] * (K + 1); spl[0] = 1; spl[K] = n + 1
    c_lo, c_hi = bounds
    for k in range(1, K + 1):
        jhi = spl_hi[k]
        jlo = max(spl[k - 1], spl_lo[k])
        while jlo <= jhi:
            jp = fld2(jlo + jhi)
            c = fn(spl[k - 1], jp, k)
            if c_lo <= c < c_hi:
                chk = True
                spl[k] = jp
                for kk in range(k + 1, K):
                    spl[kk] = (search_flip if flip else search)(spl[kk - 1], spl_lo[kk], spl_hi[kk], kk, c)
                    if (not flip and spl[kk] < spl[kk - 1]) or (flip and spl[kk] > n + 1):
                        chk = False
                        for t in range(kk, K):
                            spl[t] = spl[kk - 1] if not flip else n + 1
                        break
                ok = chk and fn(spl[K - 1], spl[K], K) <= c
                if ok:
                    c_hi = c
                    if not flip:
                        jhi = jp - 1; spl_hi = list(spl)
                    else:
                        jlo = jp + 1; spl_lo = list(spl)
                else:
                    c_lo = c
                    if not flip:
                        jlo = jp + 1; spl_lo = list(spl)
                    else:
                        jhi = jp - 1; spl_hi = list(spl)
            elif c >= c_hi:
                if not flip:
                    jhi = jp - 1
                else:
                    jlo = jp + 1
            else:
                if not flip:
                    jlo = jp + 1
                else:
                    jhi = jp - 1
        if not flip:
            if jhi < spl[k - 1]:
                break
            spl[k] = jhi
        else:
            if jlo > n + 1:
                break
            spl[k] = jlo
    return spl_lo if flip else spl_hi


def py_lazy(A, K, f, bounds, eps):
    """LazyBisectCostBottleneckSplitter.jl:140-258 transcribed independently (hst / cch arrays and all)."""
    n, m = A.n, A.m
    pos, idx = A.colptr, A.rowval
    spl = [0] * (K + 1); spl[0] = 1
    spl_hi = [n + 1] * (K + 1); spl_hi[0] = 1
    hst = [0] * (m + 1)
    cch = [0] * (A.nnz + 1)
    c_lo, c_hi = float(bounds[0]), float(bounds[1])
    for k in range(1, K + 1):
        c_lo = max(c_lo, float(f(0, 0, 0, k)))
    state = {"first": True}

    def probe(c):
        first = state["first"]
        spl[0] = 1
        j, k, nv, npin, nn = 1, 1, 0, 0, 0
        for jp in range(1, n + 1):
            nv += 1
            npin += int(pos[jp] - pos[jp - 1])
            for q in range(int(pos[jp - 1]), int(pos[jp])):
                if first:
                    i = int(idx[q - 1])
                    if hst[i] < j:
                        nn += 1
                    cch[q] = hst[i]
                    hst[i] = jp
                elif cch[q] < j:
                    nn += 1
            while (k < K or not first) and f(nv, npin, nn, k) > c:
                if not first and k == K:
                    return False
                spl[k] = jp
                j = jp
                k += 1
                nv = 1
                npin = nn = int(pos[jp] - pos[jp - 1])
        res = True
        if first:
            res = k < K or f(nv, npin, nn, K) <= c
        while k <= K:
            spl[k] = n + 1
            k += 1
        return res

    while c_lo * (1 + eps) < c_hi:
        c = (c_lo + c_hi) / 2
        if probe(c):
            c_hi = c
            spl_hi = list(spl)
        else:
            c_lo = c
        state["first"] = False
    return spl_hi


def _mats():
    return small_matrices(40, trials=1) + [golden_matrices()["HB/can_292"]]


def test_bisect_index_is_exact_and_matches_independent_statement(orc):
    """test_Partitioners.jl:101 pins BisectIndex at eps = 0 against ReferenceBottleneckSplitter."""
    rng = np.random.default_rng(7)
    for A in _mats():
        D = dense_mask(A); T = net_table(D); S = selfnet_table(D)
        for K in (1, 2, 3, 4, 8):
            if A.n > 100 and K > 4:
                continue
            for f in (cp.AffineWorkModel(0, 10, 1), cp.AffineConnectivityModel(0, 3, 1, 3),
                      cp.AffineConnectivityModel(0, 3, 1, 3, alpha_k=rng.integers(1, 11, K).tolist())):
                ref = cp.partition_stripe(A, K, cp.ReferenceBottleneckSplitter(f), backend=orc)
                c = cp.bottleneck_value(A, ref, f, backend=orc)
                got = cp.partition_stripe(A, K, cp.BisectIndexBottleneckSplitter(f), backend=orc)
                s = got.spl
                assert np.all(np.diff(s) >= 0) and s[0] == 1 and s[-1] == A.n + 1 and got.K == K
                assert cp.bottleneck_value(A, got, f, backend=orc) == c
                if A.n <= 20 and f.alpha_k is None:
                    fn = lambda j, jp, k: model_value(f, D, A.colptr, T, S, j, jp, k)
                    bounds = cp.bound_stripe(A, K, f, backend=orc)
                    assert py_bisect_index(A, K, f, fn, bounds, 0) == s.tolist()


def test_flip_bisect_index_on_decreasing_costs(orc):
    """test_Partitioners.jl:116-152 (Funky models: per-part alpha, negative betas)."""
    rng = np.random.default_rng(8)
    for A in small_matrices(41, trials=1):
        D = dense_mask(A); T = net_table(D); S = selfnet_table(D)
        for K in (1, 2, 3, 4):
            base = 1 + A.nnz + 3 * A.n + 3 * A.m
            f = cp.AffineConnectivityModel(0, -3, -1, -3, alpha_k=(base + rng.integers(1, 11, K)).tolist())
            ref = cp.partition_stripe(A, K, cp.ReferenceBottleneckSplitter(f), backend=orc)
            c = cp.bottleneck_value(A, ref, f, backend=orc)
            # the reference's own bound_stripe asserts beta >= 0 (ConnectivityCosts.jl:30-32), so its test can only
            # run Flip methods on models whose bound_stripe is defined differently (Funky*); with our marshalled
            # model the entry reports the failed assertion
            try:
                got = cp.partition_stripe(A, K, cp.FlipBisectIndexBottleneckSplitter(f), backend=orc)
            except AssertionError:
                continue
            assert cp.bottleneck_value(A, got, f, backend=orc) == c


def test_flip_bisect_index_matches_independent_statement(orc):
    """Flip variant on a monotone model: not meaningful as a partitioner, but the control flow is fully defined and
    must match the independent statement."""
    for A in small_matrices(42, trials=1):
        if A.n > 20:
            continue
        D = dense_mask(A); T = net_table(D); S = selfnet_table(D)
        for K in (1, 2, 3, 4):
            for f in (cp.AffineWorkModel(0, 10, 1), cp.AffineConnectivityModel(0, 3, 1, 3)):
                fn = lambda j, jp, k: model_value(f, D, A.colptr, T, S, j, jp, k)
                bounds = cp.bound_stripe(A, K, f, backend=orc)
                got = cp.partition_stripe(A, K, cp.FlipBisectIndexBottleneckSplitter(f), backend=orc)
                assert py_bisect_index(A, K, f, fn, bounds, 1) == got.spl.tolist()


def test_lazy_bisect_cost(orc):
    """test_Partitioners.jl:104-105: LazyBisectCost within (1 + eps) of the optimum, for connectivity models."""
    rng = np.random.default_rng(9)
    for A in _mats():
        for K in (1, 2, 3, 4, 8):
            if A.n > 100 and K > 4:
                continue
            for f in (cp.AffineConnectivityModel(0, 3, 1, 3), cp.AffineConnectivityModel(0, 0, 0, 1),
                      cp.AffineConnectivityModel(0, 3, 1, 3, alpha_k=rng.integers(1, 11, K).tolist()),
                      cp.AffineConnectivityModel(0.0, 3.0, 1.0, 3.0)):
                ref = cp.partition_stripe(A, K, cp.ReferenceBottleneckSplitter(f), backend=orc)
                c = cp.bottleneck_value(A, ref, f, backend=orc)
                bounds = cp.bound_stripe(A, K, f, backend=orc)
                for eps in (0.1, 0.01):
                    got = cp.partition_stripe(A, K, cp.LazyBisectCostBottleneckSplitter(f, eps), backend=orc)
                    s = got.spl
                    assert np.all(np.diff(s) >= 0) and s[0] == 1 and s[-1] == A.n + 1 and got.K == K
                    assert cp.bottleneck_value(A, got, f, backend=orc) <= c * (1 + eps)
                    if A.n <= 20:
                        assert py_lazy(A, K, f, bounds, eps) == s.tolist()


def test_lazy_rejects_non_connectivity_models(orc):
    """Work models reach the generic method whose g() asserts false (LazyBisectCostBottleneckSplitter.jl:486-501)."""
    A = small_matrices(43, trials=1)[3]
    try:
        cp.partition_stripe(A, 2, cp.LazyBisectCostBottleneckSplitter(cp.AffineWorkModel(0, 10, 1), 0.1), backend=orc)
    except AssertionError:
        return
    raise AssertionError("expected the reference's assertion")


# ------------------------------------------------------------------ ConcaveTotalChunker / ConcaveTotalSplitter
def _approx(a, b):
    return abs(a - b) <= 1e-9 * max(1.0, abs(a), abs(b))


def _widths_ok(A, Phi, w, w_max):
    """every part satisfies the AffineWorkModel weight constraint"""
    s = Phi.spl
    return all(w(int(s[k + 1] - s[k]), int(A.colptr[s[k + 1] - 1] - A.colptr[s[k] - 1])) <= w_max for k in range(Phi.K))


def test_power_work_model_values(orc):
    """ConvexWorkModel / ConcaveWorkModel of test_Partitioners.jl:54-74 through the oracle's cost oracle."""
    A = golden_matrices()["HB/can_292"]
    for f in (cp.ConcaveWorkModel(0.0, 0, 1), cp.ConcaveWorkModel(-0.7, 1, 0), cp.ConvexWorkModel(0.0, 0, 1), cp.ConvexWorkModel(-0.7, 0, 1)):
        ocl = cp.oracle_stripe(cp.RandomHint(), f, A, backend=orc)
        for (j, jp) in ((1, 1), (1, 2), (3, 40), (1, A.n + 1), (100, 200)):
            want = f(jp - j, int(A.colptr[jp - 1] - A.colptr[j - 1]))
            got = ocl(j, jp)
            assert got == want if f.gamma == 2.0 else _approx(got, want)


def test_concave_total_splitter_is_optimal(orc):
    """test_Partitioners.jl:201-222: same total value as ReferenceTotalSplitter on concave costs."""
    w = cp.AffineWorkModel(0, 1, 0)
    for A in small_matrices(50, trials=1) + [golden_matrices()["HB/can_292"]]:
        for K in (1, 2, 3, 4, 8):
            if A.n > 100 and K > 4:
                continue
            for f in (cp.ConcaveWorkModel(0, 0, 1), cp.AffineWorkModel(0, 0, 0), cp.AffineWorkModel(2, 3, 1),
                      cp.ConstrainedCost(cp.ConcaveWorkModel(0, 1, 0), w, 2), cp.ConstrainedCost(cp.ConcaveWorkModel(0, 1, 0), w, 4),
                      cp.ConstrainedCost(cp.ConcaveWorkModel(0, 0, 1), w, 8)):
                ref = cp.partition_stripe(A, K, cp.ReferenceTotalSplitter(f), backend=orc)
                got = cp.partition_stripe(A, K, cp.ConcaveTotalSplitter(f), backend=orc)
                s = got.spl
                assert np.all(np.diff(s) >= 0) and s[0] == 1 and s[-1] == A.n + 1 and got.K == K
                if isinstance(f, cp.ConstrainedCost):
                    feas_ref, feas_got = _widths_ok(A, ref, w, f.w_max), _widths_ok(A, got, w, f.w_max)
                    assert feas_ref == feas_got                   # Extended costs: infinity == infinity
                    if not feas_ref:
                        continue
                assert _approx(cp.total_value(A, got, f, backend=orc), cp.total_value(A, ref, f, backend=orc)), (A, K)


def test_concave_total_chunker_is_optimal(orc):
    """test_Partitioners.jl:278-299."""
    w = cp.AffineWorkModel(0, 1, 0)
    for A in small_matrices(51, trials=1) + [golden_matrices()["HB/can_292"]]:
        if A.n < 1:
            continue
        for f in (cp.ConcaveWorkModel(0.0, 0, 1), cp.ConcaveWorkModel(-0.7, 0, 1), cp.AffineWorkModel(0, 0, 0), cp.AffineWorkModel(-2, 3, 1),
                  cp.ConstrainedCost(cp.ConcaveWorkModel(0, 1, 0), w, 2), cp.ConstrainedCost(cp.ConcaveWorkModel(0, 1, 0), w, 4),
                  cp.ConstrainedCost(cp.ConcaveWorkModel(0, 0, 1), w, 8)):
            ref = cp.pack_stripe(A, cp.ReferenceTotalChunker(f), backend=orc)
            got = cp.pack_stripe(A, cp.ConcaveTotalChunker(f), backend=orc)
            s = got.spl
            assert np.all(np.diff(s) >= 0) and s[0] == 1 and s[-1] == A.n + 1
            if isinstance(f, cp.ConstrainedCost):
                assert _widths_ok(A, got, w, f.w_max) and _widths_ok(A, ref, w, f.w_max)
            assert _approx(cp.total_value(A, got, f, backend=orc), cp.total_value(A, ref, f, backend=orc)), (A, f)


# ------------------------------------------------------------------ SURVEY 8(f) row 4, first block: adjointpattern
def test_adjointpattern_oracle(orc):
    """util.jl:67-95: the transposed pattern with rows ascending in every column; pinned against scipy's transpose."""
    import scipy.sparse as sp
    rng = np.random.default_rng(60)
    for A in small_matrices(60, trials=1) + [golden_matrices()["LPnetlib/lp_blend"]]:
        T = cp.adjointpattern(A, backend=orc)
        assert T.shape == (A.n, A.m) and T.nnz == A.nnz
        S = sp.csc_matrix((np.ones(A.nnz), A.rowval - 1, A.colptr - 1), shape=(A.m, A.n)).T.tocsc()
        S.sort_indices()
        assert np.array_equal(T.colptr, S.indptr + 1) and np.array_equal(T.rowval, S.indices + 1)
        TT = cp.adjointpattern(T, backend=orc)
        assert np.array_equal(TT.colptr, A.colptr) and np.array_equal(TT.rowval, A.rowval)
