"""GPU parity for SURVEY 8(f)-4: the primary / secondary connectivity costs through the C ABI (oracle values, objectives,
bounds, the K-part DP) against the CPU oracle, and the alternating plaid partitioner built on them."""
import numpy as np
import pytest

from util import cp, sprand, suitesparse_shaped

pytestmark = pytest.mark.gpu


def rand_split(rng, n, K):
    return cp.SplitPartition(K, np.concatenate([[1], np.sort(rng.integers(1, n + 2, K - 1)), [n + 1]]))


def test_primary_secondary_costs_match_oracle(hip, orc):
    rng = np.random.default_rng(80)
    mats = [sprand(m, n, p, rng) for (m, n, p) in ((1, 1, 0.5), (3, 5, 0.4), (8, 8, 0.3), (20, 33, 0.15), (64, 40, 0.1))] + [suitesparse_shaped(600, 5, 3)]
    for A in mats:
        for K in (1, 2, 4):
            Pi = rand_split(rng, A.m, K); Phi = rand_split(rng, A.n, K)
            Pm = cp.MapPartition(K, rng.integers(1, K + 1, A.m))
            adjA = cp.adjointpattern(A, backend=hip)
            for prm in ((0, 0, 0, 0, 1), (2, 3, 1, 3, 6), (0.5, 1.0, 1.0, 2.0, 4.0)):
                comm = cp.AffinePrimaryConnectivityModel(*prm); local = cp.AffineSecondaryConnectivityModel(*prm)
                j = rng.integers(1, A.n + 2, 12); jp = np.array([rng.integers(a, A.n + 2) for a in j]); k = rng.integers(1, K + 1, 12)
                for P in (Pi, Pm):
                    a = cp.oracle_stripe(cp.StepHint(), comm, A, P, backend=hip)(j, jp, k)
                    b = cp.oracle_stripe(cp.StepHint(), comm, A, P, backend=orc)(j, jp, k)
                    assert np.array_equal(a, b)
                    for val in (cp.total_value, cp.bottleneck_value):
                        assert val(A, Phi, comm, P, backend=hip) == val(A, Phi, comm, P, backend=orc)
                assert cp.bound_stripe(A, K, comm, Pi, backend=hip) == cp.bound_stripe(A, K, comm, Pi, backend=orc)
                # secondary model on the adjoint (rows of the adjoint = columns of A are owned by Phi)
                i = rng.integers(1, A.m + 2, 12); ip = np.array([rng.integers(a, A.m + 2) for a in i])
                a = cp.oracle_stripe(cp.StepHint(), local, adjA, Phi, backend=hip)(i, ip, k)
                b = cp.oracle_stripe(cp.StepHint(), local, adjA, Phi, backend=orc)(i, ip, k)
                assert np.array_equal(a, b)
                for val in (cp.total_value, cp.bottleneck_value):
                    v = val(adjA, Pi, local, Phi, backend=hip)
                    assert v == val(adjA, Pi, local, Phi, backend=orc) == val(A, Phi, comm, Pi, backend=hip)      # test_Costs.jl:77
                assert cp.bound_stripe(adjA, K, local, Phi, backend=hip) == cp.bound_stripe(adjA, K, local, Phi, backend=orc)


def test_dp_on_plaid_costs_matches_oracle(hip, orc):
    rng = np.random.default_rng(81)
    mats = [sprand(m, n, p, rng) for (m, n, p) in ((3, 5, 0.4), (8, 8, 0.3), (20, 33, 0.15), (64, 70, 0.08))] + [suitesparse_shaped(400, 5, 4)]
    for A in mats:
        for K in (1, 2, 3, 5):
            Pi = rand_split(rng, A.m, K); Phi = rand_split(rng, A.n, K)
            adjA = cp.adjointpattern(A, backend=hip)
            for prm in ((0, 2, 1, 3, 6), (0, 0, 0, 0, 1), (1.0, 0.5, 1.0, 2.0, 4.5)):
                comm = cp.AffinePrimaryConnectivityModel(*prm); local = cp.AffineSecondaryConnectivityModel(*prm)
                for meth in (cp.DynamicBottleneckSplitter, cp.DynamicTotalSplitter):
                    got = cp.partition_stripe(A, K, meth(comm), Pi, backend=hip)
                    want = cp.partition_stripe(A, K, meth(comm), Pi, backend=orc)
                    assert got == want, (A, K, prm, meth.__name__, "primary")
                    got = cp.partition_stripe(adjA, K, meth(local), Phi, backend=hip)
                    want = cp.partition_stripe(adjA, K, meth(local), Phi, backend=orc)
                    assert got == want, (A, K, prm, meth.__name__, "secondary")
