"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.
Bit-exact split indices and bit-exact DP tables (values and argmins, ties included)."""
import numpy as np
import pytest

from util import cp, sprand, golden_matrices, suitesparse_shaped, banded

pytestmark = pytest.mark.gpu


def mats():
    rng = np.random.default_rng(0xDEADBEEF)
    out = [sprand(m, n, p, rng) for (m, n, p) in [(1, 1, 0.5), (3, 2, 0.5), (5, 7, 0.4), (8, 16, 0.3), (10, 23, 0.2),
                                                  (6, 33, 0.3), (20, 40, 0.1), (3, 12, 0.6), (40, 100, 0.05), (9, 64, 0.2),
                                                  (9, 65, 0.2), (9, 63, 0.2), (4, 8, 0.0)]]
    out += list(golden_matrices().values())
    out += [suitesparse_shaped(1000, 6, 3), banded(777, 4, 0.5, 9)]
    return out


MODELS = [cp.AffineConnectivityModel(0, 0, 0, 1), cp.AffineConnectivityModel(0, 10, 1, 100),
          cp.AffineConnectivityModel(0.0, 0.0, 0.0, 1.0), cp.AffineConnectivityModel(2, -3, 1, 3),
          cp.AffineWorkModel(0, 10, 1),
          # hyperedge-cut costs in the inverse-Monge class (b_cut >= 0, b_self <= b_cut): fast scheme with two counts
          cp.AffineHyperedgeCutModel(0, 0, 0, 0, 1), cp.AffineHyperedgeCutModel(0, 2, 1, 1, 3),
          cp.AffineHyperedgeCutModel(0.0, 0.0, 0.0, -1.0, 0.0),
          # outside the class (b_self > b_cut): must take the general sweep and still match
          cp.AffineHyperedgeCutModel(0, 0, 0, 1, 0)]


def test_link_array_matches_reference_sweep(hip, orc):
    for A in mats():
        assert np.array_equal(hip.link_array(A), orc.link_array(A))


@pytest.mark.parametrize("mi", range(len(MODELS)))
def test_total_splitter_tables_bit_exact(hip, orc, mi):
    mdl = MODELS[mi]
    for A in mats():
        for K in (1, 2, 5):
            mm = mdl.marshal()
            rc1, p1, c1 = hip.dynamic_tables(A, K, 0, mm, None)
            rc2, p2, c2 = orc.dynamic_tables(A, K, 0, mm, None)
            assert rc1 == 0 and rc2 == 0, hip.last_error()
            assert np.array_equal(p1, p2), (A, K, mi)
            assert np.array_equal(c1, c2), (A, K, mi)
            got = cp.partition_stripe(A, K, cp.DynamicTotalSplitter(mdl), backend=hip)
            want = cp.partition_stripe(A, K, cp.DynamicTotalSplitter(mdl), backend=orc)
            assert got == want
            got2 = cp.partition_stripe(A, K, cp.DynamicTotalChunker(mdl), backend=hip)
            assert got2 == want


def test_fast_path_equals_general_sweep(hip, orc):
    """The O(n log^2 n) scheme and the literal O(n^2) device sweep agree with the oracle on a mid-size input."""
    A = suitesparse_shaped(3000, 8, 11)
    for mdl in (cp.AffineConnectivityModel(0, 0, 0, 1), cp.AffineConnectivityModel(0, 10, 1, 100),
                cp.AffineHyperedgeCutModel(0, 0, 0, 0, 1), cp.AffineHyperedgeCutModel(0, 10, 1, 30, 100)):
        K = 6
        want = cp.partition_stripe(A, K, cp.DynamicTotalSplitter(mdl), backend=orc)
        fast = cp.partition_stripe(A, K, cp.DynamicTotalSplitter(mdl), backend=hip)
        hip.set_option("force_brute", 1)
        try:
            brute = cp.partition_stripe(A, K, cp.DynamicTotalSplitter(mdl), backend=hip)
        finally:
            hip.set_option("force_brute", 0)
        assert fast == want and brute == want
        assert cp.total_value(A, fast, mdl, backend=hip) == cp.total_value(A, want, mdl, backend=orc)


@pytest.mark.parametrize("g", ["sum", "max"])
def test_general_sweep_all_models(hip, orc, g):
    """Bottleneck objective, hyperedge cut, per-part alpha[k], decreasing costs: general device sweep."""
    rng = np.random.default_rng(5)
    ms = [sprand(8, 16, 0.3, rng), sprand(20, 40, 0.1, rng), golden_matrices()["LPnetlib/lpi_itest6"],
          golden_matrices()["Pajek/GD99_c"], suitesparse_shaped(300, 5, 2)]
    for A in ms:
        for K in (1, 2, 3, 4, 8):
            base = 1 + A.nnz + 3 * A.n + 3 * A.m
            models = [cp.AffineConnectivityModel(0, 3, 1, 3), cp.AffineWorkModel(0, 10, 1),
                      cp.AffineHyperedgeCutModel(0, 0, 0, 0, 1), cp.AffineHyperedgeCutModel(0, 1, 1, 1, 3),
                      cp.AffineHyperedgeCutModel(0, 0, 0, 1, 0),
                      cp.AffineConnectivityModel(0, 3, 1, 3, alpha_k=rng.integers(1, 11, K).tolist()),
                      cp.AffineConnectivityModel(0, -3, -1, -3, alpha_k=(base + rng.integers(1, 11, K)).tolist()),
                      cp.AffineConnectivityModel(-0.5, 0.25, 0.0, 1.5)]
            for mdl in models:
                meth = (cp.DynamicTotalSplitter if g == "sum" else cp.DynamicBottleneckSplitter)(mdl)
                got = cp.partition_stripe(A, K, meth, backend=hip)
                want = cp.partition_stripe(A, K, meth, backend=orc)
                assert got == want, (A, K, mdl.kind, g)
                f = cp.total_value if g == "sum" else cp.bottleneck_value
                a, b = f(A, got, mdl, backend=hip), f(A, want, mdl, backend=orc)
                assert a == b or abs(a - b) <= 1e-12 * abs(b)        # Float64 totals: 1e-12 relative


def test_objective_and_bounds(hip, orc):
    rng = np.random.default_rng(8)
    for A in [sprand(20, 40, 0.1, rng), golden_matrices()["HB/can_292"]]:
        for K in (1, 3, 8):
            Phi = cp.partition_stripe(A, K, cp.EquiSplitter())
            for mdl in (cp.AffineWorkModel(0, 10, 1), cp.AffineConnectivityModel(0, 10, 1, 100),
                        cp.AffineHyperedgeCutModel(0, 0, 0, 1, 1), cp.ColumnBlockComponentCostModel(3, lambda w: 1 + w)):
                for f in (cp.total_value, cp.bottleneck_value):
                    assert f(A, Phi, mdl, backend=hip) == f(A, Phi, mdl, backend=orc)
            for mdl in (cp.AffineWorkModel(0, 10, 1), cp.AffineConnectivityModel(0, 10, 1, 100)):
                assert cp.bound_stripe(A, K, mdl, backend=hip) == cp.bound_stripe(A, K, mdl, backend=orc)


def test_no_read_before_write_in_round_buffers(hip, orc):
    """The per-round work buffers (step counts, tile partials) are recycled allocations: poison them before every round
    (cp_set_option("dbg", 128 [+256]): all-0x7F / all-zero bytes) and demand identical tables -- a kernel that read a
    cell another kernel was supposed to write shows up as a mismatch for one of the two patterns."""
    A = banded(777, 4, 0.5, 9)
    B = suitesparse_shaped(3000, 6, 5)
    for mdl in (MODELS[0], MODELS[1], MODELS[6]):
        mm = mdl.marshal()
        for M_ in (A, B):
            rc2, p2, c2 = orc.dynamic_tables(M_, 5, 0, mm, None)
            for dbg in (128, 128 + 256, 64, 64 + 128, 0):       # 64: long tasks stay in the flattened space (no tiles of their own)
                hip.set_option("dbg", dbg)
                try:
                    rc1, p1, c1 = hip.dynamic_tables(M_, 5, 0, mm, None)
                finally:
                    hip.set_option("dbg", 0)
                assert rc1 == 0 and np.array_equal(p1, p2) and np.array_equal(c1, c2), (dbg, M_)
            for (st, se, om) in ((0, 0, 64), (1, 64, 256), (100, 100000, 1000), (4, 64, 100)):
                hip.set_option("short_t", st); hip.set_option("short_e", se); hip.set_option("own_min", om)
                try:
                    rc1, p1, c1 = hip.dynamic_tables(M_, 5, 0, mm, None)
                finally:
                    hip.set_option("short_t", 8); hip.set_option("short_e", 64); hip.set_option("own_min", 64)
                assert rc1 == 0 and np.array_equal(p1, p2) and np.array_equal(c1, c2), ("short", st, om, M_)


# every tunable of the layer driver selects between code paths that must all produce the reference's tables:
#   nospec     1: every round waits for its exact counts / 0: grids and buffers sized from the previous layer's counts
#   gap_tau    rounds tau <= gap_tau finish whole gaps in one pass (-1: plain divide and conquer everywhere)
#   gap_min    shortest task taken by a gap pass
#   ra_cache   round A from counts computed once per partition / recomputed by every layer
#   fixed_point 1: layers after a converged layer are copied instead of recomputed
#   dbg 512    every gap tile takes the entry-by-entry path (as if it held too many specials)
LAYER_OPTIONS = [{"nospec": 1}, {"gap_tau": -1}, {"gap_tau": 0}, {"gap_tau": 3, "gap_min": 8}, {"gap_tau": 8, "gap_min": 16}, {"gap_tau": 12},
                 {"ra_cache": 0}, {"ra_cache": 0, "gap_tau": -1, "nospec": 1}, {"dbg": 512}, {"dbg": 512, "gap_tau": 7, "gap_min": 8},
                 {"rpass_small_tau": 6}, {"rpass_ch": 16}, {"rpass_small_tau": -1, "rpass_ch": 16}, {"rpass_cap": 1}, {"dbg": 524288}, {"force_max": 1000000}, {"dbg": 262144}, {"dbg": 16384}, {"setup_bs": 128}, {"rpass_small_tau": 9, "rpass_cap": 30},
                 # mispredictions: stages skipped although they have work / buffers too small -- the layer must notice and redo itself
                 {"dbg": 1024}, {"dbg": 2048}, {"dbg": 1024 + 2048, "gap_tau": -1},
                 # layers after one that reproduced its input row are copied (exact; off by default)
                 {"fixed_point": 1}]
LAYER_DEFAULTS = {"nospec": 0, "gap_tau": 6, "gap_min": 64, "ra_cache": 1, "dbg": 0, "rpass_small_tau": 4, "rpass_ch": 256, "rpass_cap": 200, "force_max": 1024, "setup_bs": 1024, "fixed_point": 0}


@pytest.mark.parametrize("oi", range(len(LAYER_OPTIONS)))
def test_layer_driver_options_bit_exact(hip, orc, oi):
    opts = LAYER_OPTIONS[oi]
    mats_ = list(golden_matrices().values())[-2:] + [suitesparse_shaped(5000, 8, 1), banded(2500, 6, 0.5, 3), suitesparse_shaped(1025, 5, 7)]
    try:
        for k, v in opts.items():
            assert hip.set_option(k, v) == 0
        for A in mats_:
            for mdl in (MODELS[0], MODELS[3], MODELS[6], MODELS[7]):
                mm = mdl.marshal()
                rc1, p1, c1 = hip.dynamic_tables(A, 5, 0, mm, None)
                rc2, p2, c2 = orc.dynamic_tables(A, 5, 0, mm, None)
                assert rc1 == 0 and rc2 == 0, hip.last_error()
                assert np.array_equal(p1, p2) and np.array_equal(c1, c2), (opts, A, mdl)
    finally:
        for k, v in LAYER_DEFAULTS.items():
            hip.set_option(k, v)


def test_plain_c_client_runs(hip):
    """The C ABI driven from a C program (examples/c_abi_demo.c) in a child process: no Python between caller and library."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "examples"), "c_abi_demo"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(root, "examples", "c_abi_demo")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout + out.stderr


def test_device_block_pool_on_and_off(hip, orc):
    """cp_set_option("pool", ...): freed device blocks of 1 MB and more are kept for the next call (csrc/core.hip dev_alloc) -- the same
    answers with the pool off, after it was emptied, and on again with recycled (non-zero) blocks"""
    A = suitesparse_shaped(60000, 8, 77)          # (large enough for blocks above the pool's 1 MB threshold)
    mdl = MODELS[1]
    want = None
    try:
        for pool in (1, 0, 1, 1):
            assert hip.set_option("pool", pool) == 0
            got = [cp.partition_stripe(A, 5, cp.DynamicTotalSplitter(cp.ConstrainedCost(mdl, cp.VertexCount(), 20000)), backend=hip),
                   cp.partition_stripe(A, 5, cp.DynamicBottleneckSplitter(mdl), backend=hip),
                   cp.partition_stripe(A, 7, cp.BisectCostBottleneckSplitter(mdl, 0.01), backend=hip)]
            if want is None:
                want = got
            assert got == want
    finally:
        hip.set_option("pool", 1)
    assert want[1] == cp.partition_stripe(A, 5, cp.DynamicBottleneckSplitter(mdl), backend=orc)
