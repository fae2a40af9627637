"""SURVEY 8(f) row 4 on the CPU oracle: Primary / Secondary connectivity costs (PrimaryConnectivityCosts.jl,
SecondaryConnectivityCosts.jl) pinned by the reference's own properties (test/test_Costs.jl:52-80) and by brute-force
definitions on dense masks."""
import numpy as np
import pytest

from util import cp, sprand, dense_mask


def brute_primary_part(D, asg, j, jp, k, mdl):
    """part k = columns j : jp-1; nets = distinct rows, local = those owned by part k (PrimaryConnectivityCosts.jl:67-74)."""
    rows = np.nonzero(D[:, j - 1:jp - 1].any(axis=1))[0] if jp > j else np.zeros(0, dtype=int)
    pins = int(D[:, j - 1:jp - 1].sum())
    local = int(np.sum(asg[rows] == k))
    return mdl(jp - j, pins, local, len(rows) - local, k)


def rand_split(rng, n, K):
    return cp.SplitPartition(K, np.concatenate([[1], np.sort(rng.integers(1, n + 2, K - 1)), [n + 1]]))


def test_primary_secondary_costs(orc):
    rng = np.random.default_rng(70)
    for m in list(range(1, 30)) + [60]:
        for K in (1, 2, 3, 4):
            n = m
            A = sprand(m, n, 0.125, rng)
            D = dense_mask(A)
            Pi = rand_split(rng, m, K)          # rows
            Phi = rand_split(rng, n, K)         # columns
            asg = cp.to_map(Pi).asg
            adjA = cp.adjointpattern(A, backend=orc)
            for prm in ((0, 0, 0, 0, 1), (0, 0, 0, 1, 1), (1, 1, 1, 1, 1), (2, 3, 1, 3, 6), (0.5, 1.0, 1.0, 2.0, 4.0)):
                comm = cp.AffinePrimaryConnectivityModel(*prm)
                local = cp.AffineSecondaryConnectivityModel(*prm)
                # oracle values against the brute-force definition
                ocl = cp.oracle_stripe(cp.StepHint(), comm, A, Pi, backend=orc)
                for _ in range(6):
                    j = int(rng.integers(1, n + 2)); jp = int(rng.integers(j, n + 2)); k = int(rng.integers(1, K + 1))
                    assert ocl(j, jp, k) == brute_primary_part(D, asg, j, jp, k, comm)
                want = max(brute_primary_part(D, asg, int(Phi.spl[k - 1]), int(Phi.spl[k]), k, comm) for k in range(1, K + 1))
                bv = cp.bottleneck_value(A, Phi, comm, Pi, backend=orc)
                assert bv == want
                tv = cp.total_value(A, Phi, comm, Pi, backend=orc)
                assert tv == sum(brute_primary_part(D, asg, int(Phi.spl[k - 1]), int(Phi.spl[k]), k, comm) for k in range(1, K + 1))
                # test_Costs.jl:69-79: the secondary model on the adjoint with the roles swapped is the same objective
                assert cp.bottleneck_value(adjA, Pi, local, Phi, backend=orc) == bv
                assert cp.total_value(adjA, Pi, local, Phi, backend=orc) == tv
                # ... and its oracle agrees part by part
                locl = cp.oracle_stripe(cp.StepHint(), local, adjA, Phi, backend=orc)
                vals = [locl(int(Pi.spl[k - 1]), int(Pi.spl[k]), k) for k in range(1, K + 1)]
                assert max(vals) == bv
                # bounds sandwich (test_Costs.jl:73-76)
                lo, hi = cp.bound_stripe(A, K, comm, Pi, backend=orc)
                assert 0 <= lo <= bv <= hi
                lo, hi = cp.bound_stripe(adjA, K, local, Phi, backend=orc)
                assert 0 <= lo <= bv <= hi


def test_dp_on_primary_and_secondary_costs(orc):
    """DynamicBottleneckSplitter / DynamicTotalSplitter with the 2-D costs: optimal against exhaustive search on tiny inputs."""
    import itertools
    rng = np.random.default_rng(71)
    for (m, n) in ((4, 5), (6, 6), (7, 5)):
        A = sprand(m, n, 0.4, rng); D = dense_mask(A)
        for K in (2, 3):
            Pi = rand_split(rng, m, K); asg = cp.to_map(Pi).asg
            comm = cp.AffinePrimaryConnectivityModel(0, 2, 1, 3, 6)
            for g, meth, val in (("max", cp.DynamicBottleneckSplitter(comm), cp.bottleneck_value), ("sum", cp.DynamicTotalSplitter(comm), cp.total_value)):
                got = cp.partition_stripe(A, K, meth, Pi, backend=orc)
                best = None
                for cuts in itertools.combinations_with_replacement(range(1, n + 2), K - 1):
                    spl = [1] + list(cuts) + [n + 1]
                    vals = [brute_primary_part(D, asg, spl[k - 1], spl[k], k, comm) for k in range(1, K + 1)]
                    v = max(vals) if g == "max" else sum(vals)
                    best = v if best is None or v < best else best
                assert val(A, got, comm, Pi, backend=orc) == best


def test_bisect_methods_on_plaid_costs(orc):
    """test_Partitioners.jl:76-113 with the primary model given Pi = EquiSplitter on the adjoint: BisectIndex is exact,
    BisectCost within (1 + eps) of the DP optimum; the secondary model likewise on the adjoint."""
    rng = np.random.default_rng(72)
    for (m, n) in ((8, 8), (12, 20), (30, 25), (64, 64)):
        A = sprand(m, n, 0.15, rng)
        adjA = cp.adjointpattern(A, backend=orc)
        for K in (1, 2, 3, 4):
            Pi = cp.partition_stripe(adjA, K, cp.EquiSplitter())
            Phi = cp.partition_stripe(A, K, cp.EquiSplitter())
            for f, X, P in ((cp.AffinePrimaryConnectivityModel(0, 2, 1, 3, 6), A, Pi), (cp.AffineSecondaryConnectivityModel(0, 2, 1, 3, 6), adjA, Phi)):
                ref = cp.partition_stripe(X, K, cp.DynamicBottleneckSplitter(f), P, backend=orc)
                c = cp.bottleneck_value(X, ref, f, P, backend=orc) if isinstance(f, cp.AffineSecondaryConnectivityModel) is False else None
                if isinstance(f, cp.AffineSecondaryConnectivityModel):
                    ocl = cp.oracle_stripe(cp.StepHint(), f, X, P, backend=orc)
                    val = lambda S: max(ocl(int(S.spl[k - 1]), int(S.spl[k]), k) for k in range(1, K + 1))
                else:
                    val = lambda S: cp.bottleneck_value(X, S, f, P, backend=orc)
                c = val(ref)
                got = cp.partition_stripe(X, K, cp.BisectIndexBottleneckSplitter(f), P, backend=orc)
                assert got.spl[0] == 1 and got.spl[-1] == X.n + 1 and np.all(np.diff(got.spl) >= 0)
                if isinstance(f, cp.AffinePrimaryConnectivityModel) and not isinstance(f, cp.AffineSecondaryConnectivityModel):
                    assert val(got) == c
                for eps in (0.1, 0.01):
                    if isinstance(f, cp.AffineSecondaryConnectivityModel):
                        continue            # decreasing in the range: the reference pairs it with the Flip variants
                    got = cp.partition_stripe(X, K, cp.BisectCostBottleneckSplitter(f, eps), P, backend=orc)
                    assert val(got) <= c * (1 + eps)


def test_flip_bisect_on_the_secondary_cost(orc):
    """test_Partitioners.jl:116-152: the secondary connectivity cost decreases with the range (more nets become local), so it
    goes with the Flip variants: FlipBisectIndex is exact, FlipBisectCost within (1 + eps)."""
    rng = np.random.default_rng(73)
    f = cp.AffineSecondaryConnectivityModel(0, 2, 1, 3, 6)
    for (m, n) in ((8, 8), (12, 20), (30, 25), (64, 64)):
        A = sprand(m, n, 0.15, rng)
        adjA = cp.adjointpattern(A, backend=orc)
        for K in (1, 2, 3, 4):
            Phi = cp.partition_stripe(A, K, cp.EquiSplitter())
            ocl = cp.oracle_stripe(cp.StepHint(), f, adjA, Phi, backend=orc)
            val = lambda S: max(ocl(int(S.spl[k - 1]), int(S.spl[k]), k) for k in range(1, K + 1))
            ref = cp.partition_stripe(adjA, K, cp.DynamicBottleneckSplitter(f), Phi, backend=orc)
            c = val(ref)
            got = cp.partition_stripe(adjA, K, cp.FlipBisectIndexBottleneckSplitter(f), Phi, backend=orc)
            assert got.spl[0] == 1 and got.spl[-1] == adjA.n + 1 and np.all(np.diff(got.spl) >= 0)
            assert val(got) == c
            for eps in (0.1, 0.01):
                got = cp.partition_stripe(adjA, K, cp.FlipBisectCostBottleneckSplitter(f, eps), Phi, backend=orc)
                assert val(got) <= c * (1 + eps)
