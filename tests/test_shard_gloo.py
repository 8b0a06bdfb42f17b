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


SHARD_WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, %r)
    import vx_import
    vx = vx_import.load()
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    # (1) the exchange a sharded proof makes once: every rank fills its own slots, the all-reduce returns the union (uint64, wrap-around)
    words = np.zeros(12, dtype=np.uint64)
    words[rank::world] = np.arange(rank, 12, world, dtype=np.uint64) + np.uint64(0xFFFFFFFF00000000)
    got = vx.shard.exchange_over(dist)(words)
    assert got.dtype == np.uint64 and (got == np.arange(12, dtype=np.uint64) + np.uint64(0xFFFFFFFF00000000)).all()
    # (2) partial blobs of different lengths -> rank 0 merges them: 3 map segments + 4 small tables, table t (bus order: segments,
    # Merkle, commitment, Ed25519, SHA-512) on rank t mod world; blob order is segments, commitment, Merkle, Ed25519, SHA-512
    S, F = 3, vx.lib.HR_FIXED
    full = {t: np.full(5 + t, 100 + t, dtype=np.uint64) for t in range(S + 4)}  # by BLOB index
    bus_of_blob = {0: 0, 1: 1, 2: 2, 3: S + 1, 4: S, 5: S + 2, 6: S + 3}
    hdr = np.zeros(F + S, dtype=np.uint64)
    hdr[0], hdr[1], hdr[2], hdr[3], hdr[16], hdr[21] = vx.lib.HR_MAGIC, 256, 100000, 100256, S, 1
    mine = [b for b in range(S + 4) if bus_of_blob[b] %% world == rank]
    h = hdr.copy()
    for b in mine:
        if b < S: h[F + b] = full[b].size
        else: h[17 + (b - S)] = full[b].size
    part = np.concatenate([h] + [full[b] for b in mine])
    merged = vx.shard.gather_and_merge(part, dist, lib=vx.lib)
    if rank == 0:
        segs, p_sha, p_tree, p_ed, p_h = vx.lib.split_blob_segments(merged)
        assert [int(p[0]) for p in segs] == [100, 101, 102] and [int(p[0]) for p in (p_sha, p_tree, p_ed, p_h)] == [103, 104, 105, 106]
        assert merged.size == F + S + sum(v.size for v in full.values())
        print("SHARD_MERGE_OK")
    else:
        assert merged is None
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


def test_sharded_proof_exchange_and_merge_two_ranks_gloo(tmp_path):
    """The two collectives of intra-proof sharding on CPU: the once-per-proof all-reduce of trace-cap slots and the
    gather + merge of the shards' partial blobs (vx_header_range_merge is host code).  The proving itself is covered on the GPU
    (tests/test_gpu_blake_air.py::test_shards_of_one_proof_merge_to_the_segmented_blob, tests/test_gpu_bench_ranks.py)."""
    script = tmp_path / "worker.py"
    script.write_text(SHARD_WORKER)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "SHARD_MERGE_OK" in out.stdout
