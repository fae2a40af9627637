"""`python bench.py --gpus N` must start N ranks itself (the driver's command shape) from a parent that
touches no GPU, and must refuse a world that is not N.  Rehearsed here without a GPU: --dry-run keeps the rendezvous, the
barrier-bracketed timed region, the MAX over ranks and the all_gather, over gloo, with a stand-in step."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(kw)
    return env


def test_self_launch_two_ranks_gloo():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run", "--backend", "gloo"],
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                 # ONE JSON line, rank 0's
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == [0, 1] and d["steps"] == 2 and d["warmup"] == 1
    assert len(set(d["pids"])) == 2 and os.getpid() not in d["pids"]      # two worker processes, neither of them the launcher's parent
    assert d["dry_run"] is True and d["value"] is None
    # the MAX over ranks: rank 1's stand-in step is the slower one (20 ms)
    assert d["ms_per_step"] >= 19.0


def test_world_size_must_equal_gpus():
    """a bench started as ONE rank with --gpus 2 (what round 2 silently ran as one GPU) must fail"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--backend", "gloo"],
                       env=_env(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "WORLD_SIZE" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_a_failing_rank_fails_the_launcher():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo"],      # gloo without --dry-run is refused by every rank
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
