#!/usr/bin/env python3
"""Writes tests/golden/circuit_blobs.npz: one small vx_header_range_prove blob (16 Ptiny headers, 6 authorities, 2 queries), the same
request with the hash-chain table in three map segments, and one
small vx_rotate_prove blob with their requests, made on the GPU -- so that the HOST verifiers of the two circuits (five and six
tables, two buses) are exercised by the CPU test tier and by tools/fuzz_verify_asan.py.  Run on a GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vx_import  # noqa: E402

vx = vx_import.load()
with vx.Context(0) as ctx:
    cfg = ctx.stark_config(num_queries=2)
    ch = vx.synth.Chain(16, profile="Ptiny", stride=512)
    sj = vx.synth.Justification(ch.target_block, ch.target_hash, n_auth=6)
    just = vx.lib.PackedJustification(sj, 8)
    out96, blob = ctx.header_range_prove(ctx.from_host(ch.headers), 512, ch.sizes, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, cfg, just=just)
    vx.lib.header_range_verify(blob, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, out96, cfg, authority_set_hash=sj.authority_set_hash, authority_set_id=sj.set_id)
    blob = blob.copy()
    _, seg3 = ctx.header_range_prove(ctx.from_host(ch.headers), 512, ch.sizes, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, cfg, just=just, n_segments=3)  # three map segments
    seg3 = seg3.copy()
    vx.lib.header_range_verify(seg3, 16, ch.trusted_block, ch.trusted_hash, ch.target_block, out96, cfg, authority_set_hash=sj.authority_set_hash, authority_set_id=sj.set_id)
    e = vx.synth.EpochEndHeader(140000, 5)
    rj = vx.synth.Justification(140000, e.hash, n_auth=7, n_signed=5, set_id=3)
    out32, rblob = ctx.rotate_prove(ctx.from_host(e.padded), e.size, 140000, 5, e.start_position, e.new_pubkeys, vx.lib.PackedJustification(rj, 12), cfg)
    vx.lib.rotate_verify(rblob, 3, rj.authority_set_hash, out32, cfg)
    path = os.path.join(ROOT, "tests", "golden", "circuit_blobs.npz")
    np.savez_compressed(path, hr_blob=np.asarray(blob), hr_blob_seg3=np.asarray(seg3), hr_out96=np.frombuffer(out96, dtype=np.uint8), hr_trusted_block=ch.trusted_block,
                        hr_trusted_hash=np.frombuffer(ch.trusted_hash, dtype=np.uint8), hr_target_block=ch.target_block,
                        hr_set_hash=np.frombuffer(sj.authority_set_hash, dtype=np.uint8), hr_set_id=sj.set_id,
                        rot_blob=np.asarray(rblob), rot_out32=np.frombuffer(out32, dtype=np.uint8), rot_set_hash=np.frombuffer(rj.authority_set_hash, dtype=np.uint8), rot_set_id=3)
    print("header_range blob", blob.size * 8, "bytes; rotate blob", rblob.size * 8, "bytes;", os.path.getsize(path), "bytes on disk")
