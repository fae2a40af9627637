"""Pins the oracle's counting structures against the brute-force definitions the reference's
own tests use (test_SparsePrefixMatrices.jl:14,28-54,74-92; test_SparseColorArrays.jl:1-11,15-40)."""
import numpy as np
import pytest

from util import cp, sprand, dense_mask, ref_dominancecount, ref_netcount, ref_selfnetcount, golden_matrices

DIMS = [1, 2, 3, 7, 8, 9]          # test_SparsePrefixMatrices.jl:24
COMBOS = [                          # (hint, kwargs) test_SparsePrefixMatrices.jl:28-41
    (0, {}), (0, {"H": 1}), (2, {"H": 1}), (2, {"H": 2}), (2, {"H": 3}), (2, {"H": 4}),
    (2, {"b": 1}), (2, {"b": 2}), (2, {"b": 3}), (2, {"b": 4}), (1, {}), (0, {"b": 1}),
]


@pytest.mark.parametrize("hint,kw", COMBOS)
def test_dominancecount_matches_bruteforce(orc, hint, kw):
    rng = np.random.default_rng(0xDEADBEEF)
    for m in DIMS:
        for n in DIMS:
            A = sprand(m, n, 0.5, rng)
            D = dense_mask(A)
            h = orc.count_build("dom", A, hint, **kw)
            for i in range(1, m + 2):
                for j in range(1, n + 2):
                    assert orc.count_step("dom", h, 3, i, 3, j) == ref_dominancecount(D, i, j), (hint, kw, m, n, i, j)
            orc.count_free("dom", h)


def test_dominancecount_step_walk(orc):
    """StepHint moves Same/Jump/Prev/Next (test_SparsePrefixMatrices.jl:74-92)."""
    rng = np.random.default_rng(1)
    m = n = 40
    A = sprand(m, n, 0.5, rng)
    D = dense_mask(A)
    h = orc.count_build("dom", A, 3)
    i, j = 1, 1
    assert orc.count_step("dom", h, 3, i, 3, j) == 0
    for _ in range(4000):
        mv = rng.integers(0, 6)
        if mv == 0 and i + 1 <= m + 1:
            i += 1; got = orc.count_step("dom", h, 1, i, 0, j)
        elif mv == 1 and i - 1 >= 1:
            i -= 1; got = orc.count_step("dom", h, 2, i, 0, j)
        elif mv == 2 and j + 1 <= n + 1:
            j += 1; got = orc.count_step("dom", h, 0, i, 1, j)
        elif mv == 3 and j - 1 >= 1:
            j -= 1; got = orc.count_step("dom", h, 0, i, 2, j)
        elif mv == 4:
            i, j = int(rng.integers(1, m + 2)), int(rng.integers(1, n + 2)); got = orc.count_step("dom", h, 3, i, 3, j)
        else:
            got = orc.count_step("dom", h, 0, i, 0, j)
        assert got == ref_dominancecount(D, i, j)
    orc.count_free("dom", h)


@pytest.mark.parametrize("hint", [0, 1, 2, 3])
def test_netcount_selfnetcount(orc, hint):
    rng = np.random.default_rng(2 + hint)
    mats = [sprand(m, n, 0.5, rng) for m in DIMS for n in DIMS]
    mats += [sprand(30, 25, 0.1, rng), sprand(25, 30, 0.05, rng)]
    for A in mats:
        D = dense_mask(A)
        net = orc.count_build("net", A, hint)
        snet = orc.count_build("selfnet", A, hint)
        for j in range(1, A.n + 2):
            for jp in range(j, A.n + 2):
                assert orc.count_step("net", net, 3, j, 3, jp) == ref_netcount(D, j, jp)
                assert orc.count_step("selfnet", snet, 3, j, 3, jp) == ref_selfnetcount(D, j, jp)
        orc.count_free("net", net); orc.count_free("selfnet", snet)


def test_netcount_step_protocol(orc):
    """The exact move sequence the DP uses: jump f(1, j'), then Next(j)."""
    rng = np.random.default_rng(5)
    for A in [sprand(20, 30, 0.2, rng), golden_matrices()["LPnetlib/lpi_itest6"]]:
        D = dense_mask(A)
        net = orc.count_build("net", A, 3)
        snet = orc.count_build("selfnet", A, 3)
        for jp in list(range(1, A.n + 2)) + [3, 1, A.n + 1, 2]:
            assert orc.count_step("net", net, 3, 1, 3, jp) == ref_netcount(D, 1, jp)
            assert orc.count_step("selfnet", snet, 3, 1, 3, jp) == ref_selfnetcount(D, 1, jp)
            for j in range(2, jp + 1):
                assert orc.count_step("net", net, 1, j, 0, jp) == ref_netcount(D, j, jp)
                assert orc.count_step("selfnet", snet, 1, j, 0, jp) == ref_selfnetcount(D, j, jp)
        # layer-1 protocol: Same(1), Next(j')
        net2 = orc.count_build("net", A, 3)
        assert orc.count_step("net", net2, 3, 1, 3, 1) == 0
        for jp in range(2, A.n + 2):
            assert orc.count_step("net", net2, 0, 1, 1, jp) == ref_netcount(D, 1, jp)
        for h in (net, snet, net2):
            orc.count_free("net", h)


def test_golden_matrices_counts(orc):
    rng = np.random.default_rng(7)
    for name, A in golden_matrices().items():
        D = dense_mask(A)
        for hint in (0, 2):
            net = orc.count_build("net", A, hint)
            snet = orc.count_build("selfnet", A, hint)
            dom = orc.count_build("dom", A, hint)
            for _ in range(200):
                j, jp = sorted(rng.integers(1, A.n + 2, 2).tolist())
                assert orc.count_step("net", net, 3, j, 3, jp) == ref_netcount(D, j, jp)
                assert orc.count_step("selfnet", snet, 3, j, 3, jp) == ref_selfnetcount(D, j, jp)
                i = int(rng.integers(1, A.m + 2))
                assert orc.count_step("dom", dom, 3, i, 3, jp) == ref_dominancecount(D, i, jp)
            orc.count_free("net", net); orc.count_free("selfnet", snet); orc.count_free("dom", dom)


def test_partwise(orc):
    """partwise(A, Pi) regroups nonzeros by row part (PartwiseCounts.jl:1-60)."""
    rng = np.random.default_rng(11)
    for m, n, K in [(8, 9, 3), (20, 15, 4), (7, 7, 1)]:
        A = sprand(m, n, 0.4, rng)
        asg = rng.integers(1, K + 1, m)
        npr, pios, prm, pos, idx = orc.partwise(A, K, asg)
        D = dense_mask(A)
        assert pos[0] == 1 and pos[-1] == A.nnz + 1
        for k in range(1, K + 1):
            cols = prm[pios[k - 1] - 1:pios[k] - 1]
            expect = [j for j in range(1, n + 1) if D[asg == k, j - 1].any()]
            assert cols.tolist() == expect
            for t, j in enumerate(cols):
                jj = pios[k - 1] + t
                rows = idx[pos[jj - 1] - 1:pos[jj] - 1]
                assert rows.tolist() == [i for i in range(1, m + 1) if D[i - 1, j - 1] and asg[i - 1] == k]
