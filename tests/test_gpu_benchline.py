"""bench.py's lines for BASELINE config 5 (DynamicTotalSplitter + AffineHyperedgeCutModel(0,0,0,0,1), K = 256, row-tiled with an
RCCL all_gather per layer) at a size the oracle reaches: run as the driver runs it (a child process), on the ONE RCCL rank a
one-GPU box has, and compared with the oracle's split vector on the same seeded matrix (tests/synth.py builds the same pattern for
numpy and torch).  The 8-rank run itself is the driver's (SCALE_rNN.json); the launcher's own protocol is rehearsed on gloo in
tests/test_bench_launcher.py."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import synth
from util import cp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0xDEADBEEF


def _bench(*argv):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_config5_tiled_on_one_rccl_rank_matches_the_oracle(orc):
    n, K = 4000, 256
    d = _bench("--config", "5", "--mode", "tiled", "--gpus", "1", "--n", str(n), "--nnz", str(10 * n), "--steps", "1", "--warmup", "1",
               "--no-cpu-baseline", "--emit-spl")
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["config"]["K"] == K and d["config"]["n"] == n
    assert "AffineHyperedgeCutModel" in d["metric"] and d["unit"] == "partitions/s" and d["value"] > 0
    assert d["roofline"]["bound"] == "hbm" and 0 <= d["roofline"]["frac"] < 1.5
    _, _, colptr, rowval = synth.suitesparse_shaped_np(n, 10, SEED + 5 - 1, nnz=10 * n)
    A = cp.SparseMatrixCSC(n, n, colptr, rowval)
    want = cp.partition_stripe(A, K, cp.DynamicTotalSplitter(cp.AffineHyperedgeCutModel(0, 0, 0, 0, 1)), backend=orc)
    assert d["check"]["spl"] == want.spl.tolist()


def test_constrained_line_matches_the_oracle_and_is_not_degenerate(orc):
    """the windowed path through the bench entry at an oracle size: a NON-degenerate split vector (the unconstrained total DP's is
    the closed form [1, n+1, ..., n+1] and says nothing)"""
    n, K = 6000, 16
    d = _bench("--config", "constrained", "--n", str(n), "--nnz", str(10 * n), "--parts", str(K), "--steps", "1", "--warmup", "1",
               "--no-cpu-baseline", "--emit-spl")
    _, _, colptr, rowval = synth.suitesparse_shaped_np(n, 10, SEED + 3 - 1, nnz=10 * n)
    A = cp.SparseMatrixCSC(n, n, colptr, rowval)
    f = cp.ConstrainedCost(cp.AffineConnectivityModel(0, 0, 0, 1), cp.VertexCount(), -(-3 * n // (2 * K)))
    want = cp.partition_stripe(A, K, cp.DynamicTotalSplitter(f), backend=orc)
    assert d["check"]["spl"] == want.spl.tolist() and len(set(want.spl.tolist())) > 2


@pytest.mark.parametrize("weight", ["width", "pins"])
def test_constrained_bottleneck_line_matches_the_oracle(orc, weight):
    """`--config constrained-bottleneck [--weight pins]` at an oracle size: the valley search with candidate limits through the bench entry"""
    n, K = 6000, 16
    d = _bench("--config", "constrained-bottleneck", "--weight", weight, "--n", str(n), "--nnz", str(10 * n), "--parts", str(K), "--steps", "1",
               "--warmup", "1", "--no-cpu-baseline", "--emit-spl")
    _, _, colptr, rowval = synth.suitesparse_shaped_np(n, 10, SEED + 3 - 1, nnz=10 * n)
    A = cp.SparseMatrixCSC(n, n, colptr, rowval)
    wgt, budget = (cp.VertexCount(), -(-3 * n // (2 * K))) if weight == "width" else (cp.AffineWorkModel(0, 0, 1), -(-3 * A.nnz // (2 * K)))
    f = cp.ConstrainedCost(cp.AffineConnectivityModel(0, 10, 1, 100), wgt, budget)
    want = cp.partition_stripe(A, K, cp.DynamicBottleneckSplitter(f), backend=orc)
    assert d["check"]["spl"] == want.spl.tolist() and len(set(want.spl.tolist())) == K + 1
    assert d["check"]["max_part_weight"] <= d["check"]["w_max"] == budget
