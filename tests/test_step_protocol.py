"""The Step protocol at the boundary (Costs.jl:174-195): Step(ocl)(Same|Next|Prev|Jump(j), ..(j'), Same(k)).
CPU: the oracle's StepHint structures stepped along random walks equal the brute-force counts (the reference's own test,
test_SparseColorArrays.jl:15-91 / test_Costs.jl).  GPU: cp_oracle_step returns the same values for the same walks, and refuses
a move that breaks its promise."""
import numpy as np
import pytest

import brute
from util import cp, sprand, golden_matrices, suitesparse_shaped


def random_walk(rng, n, steps):
    """(move_j, j, move_j', j') 1-based with j <= j'"""
    j, jp = 1, 1
    out = [(cp.Jump(j), cp.Jump(jp))]
    for _ in range(steps):
        opts = []
        if jp + 1 <= n + 1: opts.append("np")
        if jp - 1 >= j: opts.append("pp")
        if j + 1 <= jp: opts.append("nj")
        if j - 1 >= 1: opts.append("pj")
        opts += ["ss", "jump"]
        o = opts[rng.integers(len(opts))]
        if o == "np": jp += 1; out.append((cp.Same(j), cp.Next(jp)))
        elif o == "pp": jp -= 1; out.append((cp.Same(j), cp.Prev(jp)))
        elif o == "nj": j += 1; out.append((cp.Next(j), cp.Same(jp)))
        elif o == "pj": j -= 1; out.append((cp.Prev(j), cp.Same(jp)))
        elif o == "ss": out.append((cp.Same(j), cp.Same(jp)))
        else:
            j = int(rng.integers(1, n + 2)); jp = int(rng.integers(j, n + 2)); out.append((cp.Jump(j), cp.Jump(jp)))
    return out


def expected(A, mdl, walk, k):
    F = brute.cost_table(A, mdl, k)
    return np.array([F[m[0].arg - 1, m[1].arg - 1] for m in walk])


MODELS = [cp.AffineConnectivityModel(0, 10, 1, 100), cp.AffineHyperedgeCutModel(0, 2, 1, 1, 3), cp.AffineWorkModel(1, 10, 1),
          cp.AffineConnectivityModel(0, 3, 1, 3, alpha_k=[5, 1, 9])]


def test_oracle_step_walks_equal_brute_force(orc):
    rng = np.random.default_rng(12)
    for A in [sprand(8, 16, 0.3, rng), sprand(20, 40, 0.1, rng), golden_matrices()["LPnetlib/lpi_itest6"], suitesparse_shaped(120, 4, 3)]:
        for mdl in MODELS:
            walk = random_walk(rng, A.n, 400)
            k = 2
            got = cp.Step(cp.oracle_stripe(cp.StepHint(), mdl, A, backend=orc)).walk([(a, b, k) for a, b in walk])
            assert np.array_equal(got, expected(A, mdl, walk, k))


@pytest.mark.gpu
def test_gpu_step_walks_equal_the_oracle(hip, orc):
    rng = np.random.default_rng(13)
    for A in [sprand(8, 16, 0.3, rng), golden_matrices()["HB/can_292"], suitesparse_shaped(2000, 6, 3)]:
        for mdl in MODELS:
            walk = [(a, b, 3) for a, b in random_walk(rng, A.n, 600)]
            want = cp.Step(cp.oracle_stripe(cp.StepHint(), mdl, A, backend=orc)).walk(walk)
            got = cp.Step(cp.oracle_stripe(cp.StepHint(), mdl, A, backend=hip)).walk(walk)
            assert np.array_equal(got, want)
            # single calls keep the walk's position
            st = cp.Step(cp.oracle_stripe(cp.StepHint(), mdl, A, backend=hip))
            assert [st(*m) for m in walk[:25]] == want[:25].tolist()
    # a broken promise is refused (the reference's stepwise structure would silently return a wrong count)
    A = sprand(8, 16, 0.3, rng)
    st = cp.Step(cp.oracle_stripe(cp.StepHint(), MODELS[0], A, backend=hip))
    with pytest.raises(AssertionError):            # CP_EINVAL surfaces as Julia's AssertionError does in the host mirror
        st.walk([(cp.Jump(2), cp.Jump(5)), (cp.Next(7), cp.Same(5))])
