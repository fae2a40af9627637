"""The invariant behind the speculative DP layers, made checkable.

A layer of the O(n log^2 n) scheme is enqueued from the previous layer's per-round counts without host round trips; if a stage
it skipped turns out to have had work (or a buffer was too small) the layer is run again with exact counts.  Until the redo,
later rounds of the mispredicted layer read plane cells nobody wrote.  The invariant: *a layer that is NOT redone has read only
cells it wrote itself*.  cp_set_option("poison", 1) fills the planes with an out-of-range column (0x7F7F7F7F) before every
layer; the kernels that turn plane cells into addresses (task setup, leaf pass, combine) count and clamp what they read of it,
and the library fails with CP_EINTERNAL if a layer that read poison is not one of the redone ones.  ONE pass with the poison is
the test -- no repetition: results against the oracle, forced mispredictions among the option sets so that the redo path and the
poisoned reads really occur."""
import numpy as np
import pytest

from util import cp, suitesparse_shaped, banded

pytestmark = pytest.mark.gpu

DEFAULTS = {"dbg": 0, "poison": 0, "nospec": 0, "gap_tau": 6, "gap_min": 64, "leaf": 1}


def _run(hip, orc, opts):
    mats = [suitesparse_shaped(6000, 8, 21), banded(3000, 6, 0.5, 4), suitesparse_shaped(1537, 5, 9)]
    net = cp.AffineConnectivityModel(0, 10, 1, 100)
    hyp = cp.AffineHyperedgeCutModel(0, 2, 1, 1, 3)
    try:
        for k, v in opts.items():
            assert hip.set_option(k, v) == 0
        assert hip.set_option("poison", 1) == 0          # (also resets the two counters)
        for A in mats:
            n = A.n
            for K in (4, 9):
                for f in (net, hyp, cp.ConstrainedCost(net, cp.VertexCount(), -(-3 * n // (2 * K))), cp.ConstrainedCost(hyp, cp.VertexCount(), n // 3 + 70)):
                    for meth in (cp.DynamicTotalSplitter, cp.DynamicTotalChunker):
                        got = cp.partition_stripe(A, K, meth(f), backend=hip)
                        want = cp.partition_stripe(A, K, meth(f), backend=orc)
                        assert got == want, (opts, A, K, type(f).__name__, meth.__name__, hip.last_error())
        return hip.get_stat("spec_redo"), hip.get_stat("poison_hits")
    finally:
        for k, v in DEFAULTS.items():
            hip.set_option(k, v)


def test_layers_read_only_what_they_wrote(hip, orc):
    """defaults: speculative layers as production runs them"""
    redo, hits = _run(hip, orc, {})
    assert hits == 0 or redo > 0            # (a poisoned read can only come from a layer that was redone; the library checked each layer)


def test_forced_mispredictions_are_redone_not_trusted(hip, orc):
    """dbg 1024: every other round is predicted empty although it has work -- the stage is skipped, later rounds read cells nobody
    wrote (poison_hits > 0), the layer is redone (spec_redo > 0) and the results are still the oracle's"""
    redo, hits = _run(hip, orc, {"dbg": 1024})
    assert redo > 0 and hits > 0


def test_undersized_buffers_are_redone(hip, orc):
    """dbg 2048: the buffers sized from the prediction are made too small"""
    redo, _ = _run(hip, orc, {"dbg": 2048})
    assert redo > 0


def test_exact_layers_never_read_poison(hip, orc):
    """nospec: every layer waits for its exact counts -- no redo, no poisoned read"""
    redo, hits = _run(hip, orc, {"nospec": 1})
    assert redo == 0 and hits == 0


def test_without_leaf_pass_and_without_gap_passes(hip, orc):
    for opts in ({"leaf": 0}, {"gap_tau": -1}, {"leaf": 0, "gap_tau": -1, "dbg": 1024}):
        _run(hip, orc, opts)
