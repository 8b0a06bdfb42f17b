"""Multi-rank rehearsal with REAL proofs on one GPU: `bench.py --gpus 2` starts two ranks as child processes
(nothing re-exec'd); with VX_BENCH_BACKEND=gloo + VX_BENCH_DEVICE=0 both prove on device 0 and exchange their blobs through the
same barrier / gather / max-reduce control flow the driver's RCCL runs take (RCCL itself needs one GPU per rank: its path has
only ever run at world = 1 here -- no scaling curve is measured by this test).  Rank 0 verifies every gathered blob against the
input of the rank it came from."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_prove_distinct_inputs_and_gather():
    env = dict(os.environ, VX_BENCH_BACKEND="gloo", VX_BENCH_DEVICE="0", OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--inflight", "2", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak"
    assert line["gather"].startswith("torch.distributed.gather (gloo)")
    assert line["gathered_blobs"] == {"verified": 2, "distinct": 2}
    assert line["value"] > 0 and line["metric"] == "header_range_256 proofs/sec"
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):  # keep the line of the rehearsal (copied to profiles/ by hand)
        json.dump(line, open(os.path.join(out, "r03_bench_2ranks_gloo_1gpu.json"), "w"), indent=1)


def test_one_proof_sharded_over_two_ranks():
    """SURVEY 8 f2 across PROCESSES: `bench.py --gpus 2 --shard-proof 4` -- two ranks (here both on device 0, gloo) prove ONE
    header_range_256 input together: 4 map segments + the four small tables, table t on rank t mod 2; they all-reduce their trace
    caps once per proof, rank 0 gathers and merges the partial blobs and the product verifier accepts the result."""
    env = dict(os.environ, VX_BENCH_BACKEND="gloo", VX_BENCH_DEVICE="0", OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--shard-proof", "4", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["verified"] is True
    assert line["config"]["map_segments"] == 4 and line["value"] > 0
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        json.dump(line, open(os.path.join(out, "r03_bench_shard_proof_2ranks_gloo_1gpu.json"), "w"), indent=1)
