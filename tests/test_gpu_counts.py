"""GPU parity of the counting structures (dominancecount / netcount / selfnetcount) against brute force
(test_SparsePrefixMatrices.jl:14, test_SparseColorArrays.jl:1-11) and against the CPU oracle."""
import numpy as np
import pytest

from util import (cp, sprand, dense_mask, ref_dominancecount, ref_netcount, ref_selfnetcount, golden_matrices,
                  suitesparse_shaped)

pytestmark = pytest.mark.gpu
DIMS = [1, 2, 3, 7, 8, 9]


def all_pairs(n1, n2, upper=False):
    a, b = np.meshgrid(np.arange(1, n1 + 1), np.arange(1, n2 + 1), indexing="ij")
    a, b = a.ravel(), b.ravel()
    if upper:
        k = a <= b
        a, b = a[k], b[k]
    return a.astype(np.int64), b.astype(np.int64)


@pytest.mark.parametrize("hint", [cp.NoHint(), cp.RandomHint(), cp.SparseHint(), cp.StepHint()])
def test_counts_match_bruteforce(hip, hint):
    rng = np.random.default_rng(0xDEADBEEF)
    mats = [sprand(m, n, 0.5, rng) for m in DIMS for n in DIMS] + [sprand(70, 130, 0.05, rng), sprand(64, 64, 0.2, rng)]
    for A in mats:
        D = dense_mask(A)
        C = cp.dominancecount(A, hint, backend=hip)
        i, j = all_pairs(A.m + 1, A.n + 1)
        got = C(i, j)
        want = np.array([ref_dominancecount(D, a, b) for a, b in zip(i, j)])
        assert np.array_equal(got, want)
        net = cp.netcount(A, hint, backend=hip)
        snet = cp.selfnetcount(A, hint, backend=hip)
        j, jp = all_pairs(A.n + 1, A.n + 1, upper=True)
        assert np.array_equal(net(j, jp), np.array([ref_netcount(D, a, b) for a, b in zip(j, jp)]))
        assert np.array_equal(snet(j, jp), np.array([ref_selfnetcount(D, a, b) for a, b in zip(j, jp)]))


def test_counts_match_oracle_on_larger_inputs(hip, orc):
    rng = np.random.default_rng(3)
    for A in list(golden_matrices().values()) + [suitesparse_shaped(5000, 7, 4)]:
        j = rng.integers(1, A.n + 2, 3000); jp = rng.integers(1, A.n + 2, 3000)
        j, jp = np.minimum(j, jp).astype(np.int64), np.maximum(j, jp).astype(np.int64)
        i = rng.integers(1, A.m + 2, 3000).astype(np.int64)
        for kind, f, a, b in (("net", cp.netcount, j, jp), ("selfnet", cp.selfnetcount, j, jp), ("dom", cp.dominancecount, i, jp)):
            got = f(A, backend=hip)(a, b)
            want = f(A, backend=orc)(a, b)
            assert np.array_equal(got, want), kind


def test_partwise_matches_oracle(hip, orc):
    """partwise(A, Pi) on the device == the reference's two-pass counting sort (PartwiseCounts.jl:1-60)."""
    rng = np.random.default_rng(21)
    cases = [(sprand(8, 9, 0.4, rng), 3), (sprand(20, 15, 0.3, rng), 4), (sprand(7, 7, 0.5, rng), 1), (sprand(5, 6, 0.0, rng), 2),
             (sprand(30, 40, 0.1, rng), 7), (golden_matrices()["LPnetlib/lp_blend"], 5), (suitesparse_shaped(2000, 5, 3), 16)]
    for A, K in cases:
        asg = rng.integers(1, K + 1, A.m)
        if K >= 3:
            asg[asg == 2] = 1                      # leave a part empty
        got = hip.partwise(A, K, asg)
        want = orc.partwise(A, K, asg)
        assert got[0] == want[0]
        for a, b in zip(got[1:], want[1:]):
            assert np.array_equal(a, b)


def test_dominancesum_and_rooks_match_their_definition(hip):
    """a13: dominancesum / rookcount! / rooksum! (SparsePrefixMatrices.jl:1-250, 825-1273) against the reference test's own
    definitions, ref_dominancesum(A, i, j) = sum(A[1:i-1, 1:j-1]) with UInt weights and wrap-around sums
    (test_SparsePrefixMatrices.jl:15, 41-71), on the reference's dimensions"""
    rng = np.random.default_rng(0xDEADBEEF)
    dims = list(range(1, 17)) + [31, 32, 33, 63, 64, 65]
    for m in dims:
        for n in (1, 2, 3, 7, 8, 9, 33):
            A = sprand(m, n, 0.5, rng)
            val = rng.integers(0, 2 ** 63, A.nnz, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, A.nnz).astype(np.uint64)
            D = np.zeros((m, n), dtype=np.uint64)
            cols = np.repeat(np.arange(n), np.diff(A.colptr))
            D[A.rowval - 1, cols] = val
            P = np.zeros((m + 1, n + 1), dtype=np.uint64)
            with np.errstate(over="ignore"):
                P[1:, 1:] = np.cumsum(np.cumsum(D, axis=0, dtype=np.uint64), axis=1, dtype=np.uint64)
            h, dt = hip.domsum_build(A, val)
            ii, jj = np.meshgrid(np.arange(1, m + 2), np.arange(1, n + 2), indexing="ij")
            cnt, sm = hip.wsum_query(h, dt, ii.ravel(), jj.ravel(), unsigned=True)
            hip.wsum_free(h)
            assert np.array_equal(sm.reshape(m + 1, n + 1), P), (m, n)
            Cn = np.zeros((m + 1, n + 1), dtype=np.int64); Cn[1:, 1:] = np.cumsum(np.cumsum(D != 0, axis=0), axis=1)
            assert np.array_equal(cnt.reshape(m + 1, n + 1), Cn)
        # rooks: one point per column
        N = m
        idx = rng.permutation(N) + 1
        val = rng.integers(0, 2 ** 63, N, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
        B = np.zeros((N, N), dtype=np.uint64)
        B[idx - 1, np.arange(N)] = val
        P = np.zeros((N + 1, N + 1), dtype=np.uint64)
        with np.errstate(over="ignore"):
            P[1:, 1:] = np.cumsum(np.cumsum(B, axis=0, dtype=np.uint64), axis=1, dtype=np.uint64)
        Cn = np.zeros((N + 1, N + 1), dtype=np.int64); Cn[1:, 1:] = np.cumsum(np.cumsum(B != 0, axis=0), axis=1)
        ii, jj = np.meshgrid(np.arange(1, N + 2), np.arange(1, N + 2), indexing="ij")
        for v in (val, None):
            h, dt = hip.rook_build(N, idx, v)
            cnt, sm = hip.wsum_query(h, dt, ii.ravel(), jj.ravel(), unsigned=True)
            hip.wsum_free(h)
            assert np.array_equal(cnt.reshape(N + 1, N + 1), Cn), N
            if v is not None:
                assert np.array_equal(sm.reshape(N + 1, N + 1), P), N
    # Float64 weights: equal up to rounding
    A = sprand(40, 50, 0.3, rng)
    val = rng.random(A.nnz)
    D = np.zeros((40, 50)); D[A.rowval - 1, np.repeat(np.arange(50), np.diff(A.colptr))] = val
    P = np.zeros((41, 51)); P[1:, 1:] = np.cumsum(np.cumsum(D, axis=0), axis=1)
    h, dt = hip.domsum_build(A, val)
    ii, jj = np.meshgrid(np.arange(1, 42), np.arange(1, 52), indexing="ij")
    cnt, sm = hip.wsum_query(h, dt, ii.ravel(), jj.ravel())
    hip.wsum_free(h)
    assert np.allclose(sm.reshape(41, 51), P, rtol=1e-12, atol=1e-12)
