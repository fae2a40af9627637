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
