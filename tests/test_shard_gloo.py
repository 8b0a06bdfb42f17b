"""N > 1 path on CPU: world_size-2 gloo run of the sharding + gather logic bench.py uses."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, %r)
    import vx_import
    vx = vx_import.load()
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = vx.shard.assign_inputs(5, rank, world)
    # each rank "proves" its inputs: here the public outputs of a tiny synthetic chain per input
    blob = np.concatenate([np.frombuffer(vx.synth.Chain(8, profile="Ptiny", stride=512, seed=1000 + i).expected_outputs(8), dtype=np.uint8) for i in mine[:2]])
    got = vx.shard.gather_blobs(blob, dist)
    if rank == 0:
        assert len(got) == world
        for r in range(world):
            idx = vx.shard.assign_inputs(5, r, world)[:2]
            want = b"".join(vx.synth.Chain(8, profile="Ptiny", stride=512, seed=1000 + i).expected_outputs(8) for i in idx)
            assert got[r].tobytes() == want, r
        print("GATHER_OK")
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


def test_assign_inputs_partition(vx):
    for world in (1, 2, 4, 8):
        parts = [vx.shard.assign_inputs(13, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(13))
        assert max(map(len, parts)) - min(map(len, parts)) <= 1


def test_gather_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "GATHER_OK" in out.stdout
