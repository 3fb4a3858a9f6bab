"""`python bench.py --gpus N` started as ONE plain process must start its own ranks (the driver launches `--gpus 1` that
way; a plain `--gpus 8` used to exit with rc 1 before touching a GPU).  Driven here on CPU: PGCA_BENCH_BACKEND=gloo and
--dry-launch (ranks rendezvous + one all-reduce, no model, no kernel).  Replaces Accelerate's launcher, reference
scripts/train.py:317-322."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(*extra, n=2):
    env = dict(os.environ, PGCA_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dry-launch", *extra],
                          capture_output=True, text=True, env=env, timeout=600)


def test_plain_multi_gpu_invocation_spawns_its_ranks_and_prints_one_line():
    p = run(n=3)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 3 and res["dry_launch"] is True
    assert res["process_group"] == {"world_size": 3, "backend": "gloo"}
    assert res["ranks_summed"] == 3          # every child joined the all-reduce


def test_a_failing_rank_fails_the_launch():
    p = run("--dry-fail-rank", "1")
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_a_hung_job_is_killed_at_the_limit():
    env = dict(os.environ, PGCA_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch",
                        "--launch-timeout", "0.5"], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 124 and "killed" in p.stderr


def test_single_process_dry_run_needs_no_launcher():
    p = run(n=1)
    assert p.returncode == 0, p.stderr[-2000:]
    assert json.loads(p.stdout.strip())["process_group"]["world_size"] == 1
