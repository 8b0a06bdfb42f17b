"""bench.py --gpus N is real (VERDICT r1 weak-3 / ADVICE medium): without a launcher it spawns N ranks itself, with a
launcher it refuses a WORLD_SIZE that disagrees with --gpus.  CPU only: --dry-launch joins a gloo group, no GPU work."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_gpus_flag_spawns_that_many_ranks():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--headers", "512", "--dry-launch"], capture_output=True, text=True, timeout=600, env=clean_env())
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout  # rank 0 only
    line = lines[0]
    assert line["n_gpus"] == 2 and line["gpus_arg"] == 2
    assert sorted(r["rank"] for r in line["ranks"]) == [0, 1] and sorted(r["local_rank"] for r in line["ranks"]) == [0, 1]
    assert all(r["world_size"] == 2 and r["master_addr"] == "127.0.0.1" and r["headers"] == 512 for r in line["ranks"])


def test_single_rank_dry_launch_needs_no_launcher():
    out = subprocess.run([sys.executable, BENCH, "--dry-launch"], capture_output=True, text=True, timeout=300, env=clean_env())
    assert out.returncode == 0, out.stdout + out.stderr
    assert json.loads(out.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_world_size_mismatch_is_an_error():
    env = dict(clean_env(), WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--dry-launch"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=4" in out.stderr
