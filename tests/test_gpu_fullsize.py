"""Full-size checks on the GPU (BASELINE.json configs 2-4 shapes) through size-independent properties -- the
literal oracle cannot run at these sizes:
  * structure: sorted, spl[1] == 1, spl[end] == n+1, K parts / width limits;
  * optimality of the total-cost DP: for a connectivity cost alpha=0 the optimum of sum_k f is known in closed form,
    f(1, n+1) (all modular terms telescope and sum_k nets_k >= nets(all columns) with equality for one non-empty
    part), so total_value(DP result) must equal the cost of the single part [1, n+1);  for the hyperedge-cut cost
    (b_cut only) the optimum is 0;
  * bounds sandwich c_lo <= bottleneck <= c_hi and monotonicity in eps for BisectCost;
  * DP-optimal chunking never costs more than the Convex chunker's result."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from util import cp

pytestmark = pytest.mark.gpu


def _handle(hip, n, colptr, rowval):
    return hip.csr_from_device(n, n, int(rowval.numel()), colptr.data_ptr(), rowval.data_ptr())


def test_config3_shape_total_dp_reaches_closed_form_optimum(hip):
    from bench import gen_suitesparse_shaped
    n, N = 10_000_000, 100_000_000
    colptr, rowval = gen_suitesparse_shaped(n, N, 0xDEADBEEF + 2, torch.device("cuda", 0))
    h = _handle(hip, n, colptr, rowval)
    try:
        whole = np.array([1, n + 1], dtype=np.int64)
        for mdl, K in ((cp.AffineConnectivityModel(0, 0, 0, 1), 6), (cp.AffineConnectivityModel(0, 10, 1, 100), 3),
                       (cp.AffineHyperedgeCutModel(0, 0, 0, 0, 1), 3)):
            mm = mdl.marshal()
            spl = np.zeros(K + 1, dtype=np.int64)
            assert hip.partition_dynamic(h, K, 0, 0, mm, None, None, 0, 0.0, spl) == 0, hip.last_error()
            assert spl[0] == 1 and spl[-1] == n + 1 and np.all(np.diff(spl) >= 0)
            rc, got = hip.objective(h, K, spl, mm, None, 0)
            rc2, opt = hip.objective(h, 1, whole, mm, None, 0)
            assert rc == 0 and rc2 == 0
            assert got == opt, (mdl.kind, K, got, opt)
        # Float64 twin of the headline model, AffineConnectivityModel(0.0, 0.0, 0.0, 1.0) (test/test_Partitioners.jl:176): the same split
        # vector as the Int64 model and a floating-point total within 1e-12 relative of the exact integer total (north_star's tolerance)
        K = 6
        si, sf = np.zeros(K + 1, dtype=np.int64), np.zeros(K + 1, dtype=np.int64)
        mi, mf = cp.AffineConnectivityModel(0, 0, 0, 1).marshal(), cp.AffineConnectivityModel(0.0, 0.0, 0.0, 1.0).marshal()
        assert hip.partition_dynamic(h, K, 0, 0, mi, None, None, 0, 0.0, si) == 0, hip.last_error()
        assert hip.partition_dynamic(h, K, 0, 0, mf, None, None, 0, 0.0, sf) == 0, hip.last_error()
        assert np.array_equal(si, sf)
        rc, ti = hip.objective(h, K, si, mi, None, 0)
        rc, tf = hip.objective(h, K, sf, mf, None, 0)
        assert isinstance(tf, float) and abs(tf - float(ti)) <= 1e-12 * abs(float(ti)), (ti, tf)
        # ... and of the width-constrained DP (a non-degenerate answer)
        w = -(-3 * n // (2 * K))
        wm = cp.VertexCount().marshal()
        assert hip.partition_dynamic(h, K, 0, 0, mi, None, wm, w, float(w), si) == 0, hip.last_error()
        assert hip.partition_dynamic(h, K, 0, 0, mf, None, wm, w, float(w), sf) == 0, hip.last_error()
        assert np.array_equal(si, sf) and len(set(si.tolist())) > 2
        rc, ti = hip.objective(h, K, si, mi, None, 0)
        rc, tf = hip.objective(h, K, sf, mf, None, 0)
        assert abs(tf - float(ti)) <= 1e-12 * abs(float(ti)), (ti, tf)
        # K = 1 is the identity partition
        spl = np.zeros(2, dtype=np.int64)
        assert hip.partition_dynamic(h, 1, 0, 0, cp.AffineConnectivityModel(0, 0, 0, 1).marshal(), None, None, 0, 0.0, spl) == 0
        assert spl.tolist() == [1, n + 1]
    finally:
        hip.csr_destroy(h)


def test_config2_shape_bisect_bounds_and_eps_monotone(hip):
    from bench import gen_suitesparse_shaped
    n = 1_000_000
    colptr, rowval = gen_suitesparse_shaped(n, 13 * n, 0xDEADBEEF + 1, torch.device("cuda", 0))
    h = _handle(hip, n, colptr, rowval)
    try:
        K = 32
        for mdl in (cp.AffineWorkModel(0, 10, 1), cp.AffineConnectivityModel(0, 10, 1, 100)):
            mm = mdl.marshal()
            rc, lo, hi = hip.bound_stripe(h, K, mm)
            assert rc == 0
            prev = None
            for eps in (0.1, 0.01, 0.001):
                spl = np.zeros(K + 1, dtype=np.int64)
                assert hip.partition_bisect_cost(h, K, mm, eps, 0, spl) == 0, hip.last_error()
                assert spl[0] == 1 and spl[-1] == n + 1 and np.all(np.diff(spl) >= 0)
                rc, b = hip.objective(h, K, spl, mm, None, 1)
                assert lo <= b <= hi
                if prev is not None:
                    assert b <= prev * (1 + eps * 10)      # a tighter eps never loses more than the looser tolerance
                prev = b
    finally:
        hip.csr_destroy(h)


def test_config4_shape_width_limited_chunkers(hip):
    from bench_configs import banded_dev
    n = 100_000
    colptr, rowval = banded_dev(n, 16, 0.5, 0xDEADBEEF + 4, torch.device("cuda", 0))
    h = _handle(hip, n, colptr, rowval)
    try:
        f = cp.ColumnBlockComponentCostModel(3, lambda w: 1 + w, w_table=9)
        mm, wm = f.marshal(), cp.VertexCount().marshal()
        res = {}
        for name, fn in (("dynamic", hip.pack_dynamic), ("convex", hip.pack_convex)):
            spl = np.zeros(n + 1, dtype=np.int64); Kout = np.zeros(1, dtype=np.int64)
            assert fn(h, mm, None, wm, 8, 8.0, spl, Kout) == 0, hip.last_error()
            Kc = int(Kout[0])
            s = spl[:Kc + 1]
            assert s[0] == 1 and s[-1] == n + 1 and np.all(np.diff(s) >= 1) and np.all(np.diff(s) <= 8)
            rc, tot = hip.objective(h, Kc, np.ascontiguousarray(s), mm, None, 0)
            res[name] = tot
        assert res["dynamic"] <= res["convex"]              # the DP is optimal; the stack algorithm need not be on this model
    finally:
        hip.csr_destroy(h)


def test_fast_scheme_equals_literal_device_sweep_at_2e5(hip):
    """Beyond the CPU oracle's reach: the O(n log^2 n) scheme (gap passes over gaps of hundreds of tiles, cached round A) against
    the library's own literal O(n^2) device sweep at its size limit -- complete DP tables, bit for bit (tools/check_fast_vs_sweep.py)."""
    from util import suitesparse_shaped, banded
    n, K = 200000, 4
    for A in (suitesparse_shaped(n, 10, 5), banded(n, 16, 0.5, 2)):
        for mdl in (cp.AffineConnectivityModel(0, 10, 1, 100), cp.AffineHyperedgeCutModel(0, 2, 1, 1, 3)):
            mm = mdl.marshal()
            rc1, p1, c1 = hip.dynamic_tables(A, K, 0, mm, None)
            hip.set_option("force_brute", 1)
            try:
                rc2, p2, c2 = hip.dynamic_tables(A, K, 0, mm, None)
            finally:
                hip.set_option("force_brute", 0)
            assert rc1 == 0 and rc2 == 0, hip.last_error()
            assert np.array_equal(p1, p2) and np.array_equal(c1, c2), (A, mdl)


def test_config3_shape_constrained_and_bottleneck_dps(hip):
    """n = 10^7 / nnz = 10^8: the two DPs whose answers are NOT closed-form.
      * width-constrained total DP (bin/test_table_constrained_splits.jl:28): every part within w_max, no worse than the
        full-width feasible partition, and the same split vector whatever driver options size the speculative layers;
      * bottleneck DP: its value equals the exact BisectIndex optimum, and it does not depend on the chunking of the walk."""
    from bench import gen_suitesparse_shaped
    n, N, K = 10_000_000, 100_000_000, 16
    colptr, rowval = gen_suitesparse_shaped(n, N, 0xDEADBEEF + 2, torch.device("cuda", 0))
    h = _handle(hip, n, colptr, rowval)
    try:
        mdl = cp.AffineConnectivityModel(0, 0, 0, 1)
        mm = mdl.marshal(); wm = cp.VertexCount().marshal()
        w = -(-3 * n // (2 * K))
        spl = np.zeros(K + 1, dtype=np.int64)
        assert hip.partition_dynamic(h, K, 0, 0, mm, None, wm, w, float(w), spl) == 0, hip.last_error()
        assert spl[0] == 1 and spl[-1] == n + 1 and np.all(np.diff(spl) >= 0) and int(np.diff(spl).max()) <= w
        assert len(set(spl.tolist())) > K // 2
        rc, got = hip.objective(h, K, spl, mm, None, 0)
        full = np.minimum(1 + w * np.arange(K + 1, dtype=np.int64), n + 1)
        rc, ref = hip.objective(h, K, full, mm, None, 0)
        assert got <= ref
        for opts in ({"nospec": 1}, {"gap_tau": -1}, {"dbg": 1048576}, {"dbg": 2097152}, {"force_max": 0}):      # (the last three: round A without its caches, no forced own tiles)
            for k_, v_ in opts.items():
                hip.set_option(k_, v_)
            try:
                spl2 = np.zeros(K + 1, dtype=np.int64)
                assert hip.partition_dynamic(h, K, 0, 0, mm, None, wm, w, float(w), spl2) == 0, (opts, hip.last_error())
            finally:
                hip.set_option("nospec", 0); hip.set_option("gap_tau", 6); hip.set_option("dbg", 0); hip.set_option("force_max", 1024)
            assert np.array_equal(spl, spl2), opts
        # the chunker loop order fills the same tables
        spl3 = np.zeros(K + 1, dtype=np.int64)
        assert hip.partition_dynamic(h, K, 0, 1, mm, None, wm, w, float(w), spl3) == 0
        assert np.array_equal(spl, spl3)
        net = cp.AffineConnectivityModel(0, 10, 1, 100).marshal()
        b1 = np.zeros(K + 1, dtype=np.int64); b2 = np.zeros(K + 1, dtype=np.int64); bi = np.zeros(K + 1, dtype=np.int64)
        assert hip.partition_dynamic(h, K, 1, 0, net, None, None, 0, 0.0, b1) == 0, hip.last_error()
        hip.set_option("bn_chunk", 64)
        try:
            assert hip.partition_dynamic(h, K, 1, 0, net, None, None, 0, 0.0, b2) == 0
        finally:
            hip.set_option("bn_chunk", 8)
        assert np.array_equal(b1, b2)
        assert hip.partition_bisect_index(h, K, net, 0, bi) == 0
        v1 = hip.objective(h, K, b1, net, None, 1)[1]; vi = hip.objective(h, K, bi, net, None, 1)[1]
        assert v1 == vi and len(set(b1.tolist())) > K // 2
    finally:
        hip.csr_destroy(h)


def test_one_handle_through_many_dps_equals_fresh_handles(hip):
    """State kept on a matrix handle between calls (layer work buffers, the previous layer's task counts that size the next
    one, window anchors, counters, walk hints) must never change a result: every call of a shuffled sequence of DPs -- other
    models, part counts, widths, loop orders, objectives -- on ONE handle equals the same call on a fresh handle."""
    from bench import gen_suitesparse_shaped
    n, N = 2_000_000, 20_000_000
    dev = torch.device("cuda", 0)
    colptr, rowval = gen_suitesparse_shaped(n, N, 0xDEADBEEF + 9, dev)
    net1 = cp.AffineConnectivityModel(0, 0, 0, 1).marshal()
    net2 = cp.AffineConnectivityModel(0, 10, 1, 100).marshal()
    hyp = cp.AffineHyperedgeCutModel(0, 2, 1, 1, 3).marshal()
    hypb = cp.AffineHyperedgeCutModel(0, 2, 1, 3, 1).marshal()          # (the bottleneck valley needs beta_self >= beta_cut)
    fnet = cp.AffineConnectivityModel(0.0, 3.0, 1.0, 7.0).marshal()
    wm = cp.VertexCount().marshal()

    def w_of(K, num, den):
        return -(-num * n // (den * K))
    calls = [("sum", net1, 6, 0, w_of(6, 3, 2)), ("sum", net1, 24, 0, w_of(24, 3, 2)), ("sum", net1, 5, 0, 0), ("max", net2, 12, 0, 0),
             ("sum", hyp, 6, 0, w_of(6, 3, 2)), ("sum", net1, 6, 1, w_of(6, 3, 2)), ("sum", fnet, 9, 0, w_of(9, 5, 4)), ("max", hypb, 7, 0, 0),
             ("sum", net2, 3, 0, 0), ("sum", net1, 6, 0, w_of(6, 11, 10)), ("max", fnet, 20, 0, 0), ("sum", hyp, 4, 0, 0)]

    def run(h, c):
        kind, mm, K, order, w = c
        spl = np.zeros(K + 1, dtype=np.int64)
        rc = hip.partition_dynamic(h, K, 1 if kind == "max" else 0, order, mm, None, wm if w else None, w, float(w), spl)
        assert rc == 0, (c[0], K, order, w, hip.last_error())
        return spl
    fresh = []
    for c in calls:
        h = _handle(hip, n, colptr, rowval)
        try:
            fresh.append(run(h, c))
        finally:
            hip.csr_destroy(h)
    assert sum(len(set(s.tolist())) > 3 for s in fresh) >= 6          # (the constrained and bottleneck answers are not degenerate)
    rng = np.random.default_rng(5)
    h = _handle(hip, n, colptr, rowval)
    try:
        for rep in range(3):
            for i in rng.permutation(len(calls)):
                assert np.array_equal(run(h, calls[i]), fresh[i]), (rep, i, calls[i][0], calls[i][2:])
    finally:
        hip.csr_destroy(h)
