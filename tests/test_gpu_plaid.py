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


def test_partition_plaid_alternating(hip, orc):
    """partition_plaid with the reference benchmark's models (runbenchmarks.jl:16-20: net_model, comm_model, local_model):
    the same (Pi, Phi) as the oracle, well-formed, and the later sweeps never make the communication bottleneck worse
    than the first sweep left it for the partition they start from."""
    rng = np.random.default_rng(82)
    net = cp.AffineConnectivityModel(0, 10, 1, 100)
    comm = cp.AffinePrimaryConnectivityModel(0, 10, 1, 0, 100)
    local = cp.AffineSecondaryConnectivityModel(0, 10, 1, 0, 100)
    for A in (sprand(30, 40, 0.1, rng), sprand(64, 64, 0.08, rng), suitesparse_shaped(500, 5, 9)):
        for K in (2, 4):
            for method in (cp.DisjointPartitioner(cp.DynamicBottleneckSplitter(net), cp.DynamicBottleneckSplitter(local)),
                           cp.AlternatingPartitioner(cp.DynamicBottleneckSplitter(net), cp.DynamicBottleneckSplitter(local)),
                           cp.AlternatingPartitioner(cp.DynamicBottleneckSplitter(net), cp.DynamicBottleneckSplitter(local),
                                                     cp.DynamicBottleneckSplitter(comm), cp.DynamicBottleneckSplitter(local)),
                           cp.AlternatingNetPartitioner(cp.DynamicTotalSplitter(net), cp.DynamicTotalSplitter(local), cp.DynamicTotalSplitter(comm))):
                Pi, Phi = cp.partition_plaid(A, K, method, backend=hip)
                Pi2, Phi2 = cp.partition_plaid(A, K, method, backend=orc)
                assert Pi == Pi2 and Phi == Phi2
                assert Pi.spl[0] == 1 and Pi.spl[-1] == A.m + 1 and Phi.spl[0] == 1 and Phi.spl[-1] == A.n + 1
                assert np.all(np.diff(Pi.spl) >= 0) and np.all(np.diff(Phi.spl) >= 0)
    # a sweep with the primary model given Pi is optimal for that Pi: no worse than the columns it started from
    A = suitesparse_shaped(500, 5, 9); K = 4
    Pi, Phi0 = cp.partition_plaid(A, K, cp.AlternatingPartitioner(cp.DynamicBottleneckSplitter(net), cp.DynamicBottleneckSplitter(local)), backend=hip)
    Phi1 = cp.partition_stripe(A, K, cp.DynamicBottleneckSplitter(comm), Pi, backend=hip)
    assert cp.bottleneck_value(A, Phi1, comm, Pi, backend=hip) <= cp.bottleneck_value(A, Phi0, comm, Pi, backend=hip)
    if A.m == A.n:
        P, P2 = cp.partition_plaid(A, K, cp.SymmetricPartitioner(cp.DynamicBottleneckSplitter(net)), backend=hip)
        assert P == P2 == cp.partition_stripe(A, K, cp.DynamicBottleneckSplitter(net), backend=orc)


def test_bisect_methods_on_plaid_costs_match_oracle(hip, orc):
    """BisectCost / BisectIndex (+Flip) with the primary model given Pi and the secondary model on the adjoint given Phi --
    the scalable methods of the reference's 2-D benchmark (runbenchmarks.jl:65-78): bit-exact split vectors."""
    rng = np.random.default_rng(83)
    mats = [sprand(m, n, p, rng) for (m, n, p) in ((3, 5, 0.4), (8, 8, 0.3), (20, 33, 0.15), (64, 70, 0.08))] + [suitesparse_shaped(3000, 6, 5)]
    for A in mats:
        adjA = cp.adjointpattern(A, backend=hip)
        for K in (1, 2, 4, 7):
            Pi = cp.partition_stripe(adjA, K, cp.EquiSplitter()); Phi = cp.partition_stripe(A, K, cp.EquiSplitter())
            Pm = cp.MapPartition(K, (np.arange(A.m) % K) + 1)                     # runbenchmarks.jl:67: mod1.(1:m, K)
            for prm in ((0, 10, 1, 0, 100), (0, 2, 1, 3, 6), (0.5, 1.0, 1.0, 2.0, 4.0)):
                comm = cp.AffinePrimaryConnectivityModel(*prm); local = cp.AffineSecondaryConnectivityModel(*prm)
                for P in (Pi, Pm):
                    for meth in (cp.BisectCostBottleneckSplitter(comm, 0.01), cp.BisectIndexBottleneckSplitter(comm),
                                 cp.FlipBisectIndexBottleneckSplitter(comm)):
                        got = cp.partition_stripe(A, K, meth, P, backend=hip)
                        want = cp.partition_stripe(A, K, meth, P, backend=orc)
                        assert got == want, (A, K, prm, type(meth).__name__, type(P).__name__)
                for meth in (cp.FlipBisectCostBottleneckSplitter(local, 0.01), cp.FlipBisectIndexBottleneckSplitter(local),
                             cp.BisectIndexBottleneckSplitter(local)):
                    got = cp.partition_stripe(adjA, K, meth, Phi, backend=hip)
                    want = cp.partition_stripe(adjA, K, meth, Phi, backend=orc)
                    assert got == want, (A, K, prm, type(meth).__name__, "secondary")
