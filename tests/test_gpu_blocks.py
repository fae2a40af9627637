"""GPU parity of the O(n log^2 n) DP layer at the level the closed form cannot hide.

For every model the fast path admits, the TRUE previous layer makes the diagonal candidate j = j' tie the minimum, so
ptr[j', k] = j' and the per-block arg-mins of csrc/dp_total.hip never reach cst / ptr.  Here the layer
kernel is driven directly (cp_dp_layer) with previous-layer rows that are NOT a DP row -- arbitrary, monotone, tie-heavy --
for which W[p] + f(p, r) is still inverse-Monge, and compared with brute force (tests/brute.py: counts from their
definitions, the recurrence as written):
  * the combined row: cst[r], ptr[r] (the largest minimiser over 0 <= p <= r);
  * every per-block winner: the rightmost arg-min over the Fenwick block [r_b - 2^b, r_b) of every set bit b of r
    (cp_dp_block_tables) and the net / self-net counts stored with it.
"""
import numpy as np
import pytest
import torch

import brute
from util import cp, sprand, golden_matrices, suitesparse_shaped, banded

pytestmark = pytest.mark.gpu

MODELS = [cp.AffineConnectivityModel(0, 0, 0, 1), cp.AffineConnectivityModel(1, 10, 1, 100), cp.AffineWorkModel(0, 10, 1),
          cp.AffineHyperedgeCutModel(0, 2, 1, 1, 3), cp.AffineHyperedgeCutModel(0, 0, 0, 0, 1),
          cp.AffineConnectivityModel(0.0, 0.0, 0.0, 1.0), cp.AffineHyperedgeCutModel(0.0, 1.0, 0.0, -1.0, 2.0)]

OPTIONS = [{}, {"nospec": 1}, {"gap_tau": -1}, {"gap_tau": 8, "gap_min": 8}, {"ra_cache": 0}, {"dbg": 64}, {"dbg": 512, "gap_tau": 7, "gap_min": 8},
           {"short_t": 0, "short_e": 0}, {"own_min": 1000}, {"rpass_small_tau": 6}, {"rpass_ch": 16}, {"rpass_small_tau": -1, "rpass_ch": 16}, {"rpass_cap": 1}, {"dbg": 524288}, {"force_max": 1000000}, {"dbg": 262144}, {"dbg": 16384}, {"setup_bs": 128}, {"rpass_small_tau": 9, "rpass_cap": 30}, {"dbg": 1024}, {"dbg": 2048},
           # the rounds tau < 6 as divide-and-conquer rounds (the path before the leaf pass, csrc/dp_leaf.inc), alone and with the gap passes reaching down
           {"leaf": 0}, {"leaf": 0, "gap_tau": 8, "gap_min": 8}, {"leaf": 0, "short_t": 0, "short_e": 0},
           # the gap finish with one 64-row chunk per wave (default: two), the round-A levels the leaf pass recomputes kept (dbg 33554432)
           {"gap_nr": 1}, {"gap_nr": 1, "gap_tau": 8, "gap_min": 8}, {"dbg": 33554432}]
DEFAULTS = {"nospec": 0, "gap_tau": 6, "gap_min": 64, "ra_cache": 1, "dbg": 0, "rpass_small_tau": 4, "rpass_ch": 256, "rpass_cap": 200, "force_max": 1024, "setup_bs": 1024,
            "short_t": 8, "short_e": 64, "own_min": 64, "leaf": 1, "gap_nr": 2}


def w_rows(rng, n, scale, dt):
    rows = [rng.integers(0, scale + 1, n + 1), np.sort(rng.integers(0, scale + 1, n + 1)),
            np.sort(rng.integers(0, 4 * scale + 1, n + 1))[::-1].copy(), rng.integers(0, 3, n + 1), np.zeros(n + 1, dtype=np.int64),
            # long flat stretches with a few deep wells: arg-min staircases with wide gaps (long tasks, gap passes)
            np.where(rng.random(n + 1) < 0.01, 0, scale * 8).astype(np.int64)]
    return [r.astype(dt) for r in rows]


class Tables:
    """brute-force tables of one matrix, shared by the models"""

    def __init__(self, A):
        self.A = A
        self.NT = brute.net_table(A)
        self.ST = brute.selfnet_table(A)

    def F(self, mdl, k):
        return brute.cost_table(self.A, mdl, k, self.NT, self.ST)


def check_layer(hip, A, T, mdl, W_rows, tile=None, check_blocks=True):
    n = A.n
    dev = torch.device("cuda", 0)
    mm = mdl.marshal()
    hyper = mdl.kind == cp.models.CP_MODEL_HYPEREDGE_CUT
    dt = torch.int64 if mdl.dtype == cp.models.CP_I64 else torch.float64
    lo, hi = tile if tile else (1, n + 2)
    dp = hip.dp_begin(A, 3, 0, 0, mm, lo, hi)
    moved = 0
    # the leaf pass combines its rows where it computes them; the per-block winners are stored only on request
    assert hip.set_option("block_tables", 1 if (check_blocks and tile is None) else 0) == 0
    try:
        F = T.F(mdl, 2)
        nb = max(1, int(n).bit_length())
        for W in W_rows:
            prev = torch.from_numpy(np.ascontiguousarray(W)).to(dev)
            cur = torch.zeros(n + 1, dtype=dt, device=dev)
            hip.dp_layer(dp, 2, prev.data_ptr(), cur.data_ptr())
            cst = cur.cpu().numpy()
            ptr = hip.dp_ptr_row(dp, 2, n)
            cb, pb = brute.layer(W, F)
            sl = slice(lo - 1, hi - 1)
            assert np.array_equal(ptr[sl], pb[sl] + 1), (A, mdl.kind, tile)
            assert np.array_equal(cst[sl], cb[sl].astype(cst.dtype)), (A, mdl.kind, tile)
            moved += int(np.sum(pb[sl] != np.arange(n + 1)[sl]))
            if check_blocks and tile is None:
                k, opt, nn, nl = hip.dp_block_tables(dp, n, hyper)
                assert k == nb
                ob = brute.block_argmins(W, F, nb)
                assert np.array_equal(opt - 1, ob), (A, mdl.kind)
                # the counts kept with every winner are the counts of that part
                bs, rs = np.nonzero(ob >= 0)
                assert np.array_equal(nn[bs, rs], T.NT[ob[bs, rs], rs])
                if hyper:
                    assert np.array_equal(nl[bs, rs], T.ST[ob[bs, rs], rs])
    finally:
        hip.set_option("block_tables", 0)
        hip.dp_destroy(dp)
    return moved


def small_mats():
    rng = np.random.default_rng(0xDEADBEEF)
    out = [sprand(m, n, p, rng) for (m, n, p) in [(3, 2, 0.5), (5, 7, 0.4), (8, 16, 0.3), (10, 23, 0.2), (6, 33, 0.3), (20, 40, 0.1),
                                                  (9, 64, 0.2), (9, 65, 0.2), (9, 63, 0.2), (40, 100, 0.05)]]
    out += [golden_matrices()["LPnetlib/lpi_itest6"], golden_matrices()["HB/can_292"]]
    return out


def test_block_argmins_small_matrices_all_models(hip):
    rng = np.random.default_rng(5)
    moved = 0
    for A in small_mats():
        T = Tables(A)
        for mdl in MODELS:
            dt = np.int64 if mdl.dtype == cp.models.CP_I64 else np.float64
            scale = int(abs(T.F(mdl, 2)).max()) + 1
            moved += check_layer(hip, A, T, mdl, w_rows(rng, A.n, scale, dt))
    assert moved > 2000


@pytest.mark.parametrize("oi", range(len(OPTIONS)))
def test_block_argmins_every_layer_option(hip, oi):
    """mid-size inputs (long tasks, tiles of their own, gap passes, mispredicted layers) under every driver option"""
    rng = np.random.default_rng(100 + oi)
    mats = [suitesparse_shaped(3000, 8, 1), banded(2500, 6, 0.5, 3), suitesparse_shaped(1025, 5, 7)]
    try:
        for k, v in OPTIONS[oi].items():
            assert hip.set_option(k, v) == 0
        moved = 0
        for A in mats:
            T = Tables(A)
            for mdl in (MODELS[0], MODELS[1], MODELS[3], MODELS[5]):
                dt = np.int64 if mdl.dtype == cp.models.CP_I64 else np.float64
                scale = int(abs(T.F(mdl, 2)).max()) + 1
                moved += check_layer(hip, A, T, mdl, w_rows(rng, A.n, scale, dt))
                # ... and as production runs it: no plane stores from the leaf rows, cst / ptr only
                check_layer(hip, A, T, mdl, w_rows(rng, A.n, scale, dt)[:2], check_blocks=False)
        assert moved > 10000
    finally:
        for k, v in DEFAULTS.items():
            hip.set_option(k, v)


def test_row_tiles_with_injected_rows(hip):
    """the multi-GPU row tiles compute the same rows from an injected previous layer"""
    rng = np.random.default_rng(11)
    for A in [suitesparse_shaped(2000, 6, 5), sprand(9, 65, 0.2, rng)]:
        T = Tables(A)
        n = A.n
        for mdl in (MODELS[1], MODELS[3]):
            scale = int(abs(T.F(mdl, 2)).max()) + 1
            for (lo, hi) in [(1, n // 3), (n // 3, n // 2 + 7), (n // 2 + 7, n + 2), (n + 1, n + 2)]:
                check_layer(hip, A, T, mdl, w_rows(rng, n, scale, np.int64)[:3], tile=(max(1, lo), hi))
