#!/usr/bin/env python3
"""Soak run of the product path on one GPU: many DIFFERENT header_range requests (header count 1..256, size profile, trusted block in
every SCALE compact mode, 4..300 authorities with exactly the 2/3 threshold or more signing, 1..8 map segments), each proven,
its 96 output bytes compared with the hashlib mirror, the blob accepted by the product's host verifier and refused after a
one-word mutation; every fourth request is proven twice and the bytes compared.  Prints one JSON line (profiles/r03_soak.json).
The speculative arithmetic of the NTT kernel (lazy folds with a wave-uniform exact fallback) and the multi-stream rendezvous of
the five tables see far more data here than in the test tier.
`soak.py rotate N seed` does the same for the rotate circuit: epoch-end headers with 1..300 new authorities, 0..3 logs before the
GRANDPA one, sizes from the minimum to MAX_HEADER_SIZE, justified by 4..300 current authorities.
usage: soak.py [rotate] [n_requests=60] [seed=1]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vx_import  # noqa: E402

vx = vx_import.load()
ROTATE = len(sys.argv) > 1 and sys.argv[1] == "rotate"
argv = sys.argv[2:] if ROTATE else sys.argv[1:]
N = int(argv[0]) if len(argv) > 0 else 60
rng = np.random.default_rng(int(argv[1]) if len(argv) > 1 else 1)
ctx = vx.Context(0)
cfg = ctx.stark_config()
t0 = time.time()
if ROTATE:
    stats = dict(circuit="rotate", requests=0, verified=0, mutations_refused=0, determinism_checks=0, new_authorities=0, header_bytes=0)
    for it in range(N):
        n_new = int(rng.choice([1, 2, 17, 63, 64, 65, 150, 299, 300])) if it % 3 else int(rng.integers(1, 301))
        logs_before = int(rng.integers(0, 4))
        number = int(rng.integers(1 << 14, 1 << 30))  # (the rotate statement fixes the 4-byte compact form of the epoch-end block number, as the reference does)
        probe = vx.synth.EpochEndHeader(number, n_new, seed=vx.synth.ROTATE_SEED + 7 * it, logs_before=logs_before)  # minimum size
        size = None if it % 2 else int(rng.integers(probe.size, 35841))
        e = vx.synth.EpochEndHeader(number, n_new, size=size, seed=vx.synth.ROTATE_SEED + 7 * it, logs_before=logs_before)
        n_auth = int(rng.choice([4, 9, 30, 100, 300]))
        n_signed = int(rng.integers(2 * n_auth // 3 + 1, n_auth + 1))
        set_id = int(rng.integers(0, 1 << 40))
        sj = vx.synth.Justification(e.number, e.hash, n_auth=n_auth, n_signed=n_signed, set_id=set_id, seed=vx.synth.JUST_SEED + it)
        just = vx.lib.PackedJustification(sj, 300)
        hb = ctx.from_host(e.padded)
        out32, blob = ctx.rotate_prove(hb, e.size, e.number, n_new, e.start_position, e.new_pubkeys, just, cfg)
        blob = blob.copy()
        assert out32 == e.new_authority_set_hash, f"request {it}: new authority set hash differs from the mirror"
        vx.lib.rotate_verify(blob, set_id, sj.authority_set_hash, out32, cfg)
        stats["verified"] += 1
        bad = blob.copy()
        bad[int(rng.integers(30, bad.size))] ^= np.uint64(1) << np.uint64(rng.integers(0, 64))
        try:
            vx.lib.rotate_verify(bad, set_id, sj.authority_set_hash, out32, cfg)
            raise SystemExit(f"request {it}: a mutated blob was ACCEPTED")
        except vx.VxError:
            stats["mutations_refused"] += 1
        if it % 4 == 0:
            _, again = ctx.rotate_prove(hb, e.size, e.number, n_new, e.start_position, e.new_pubkeys, just, cfg)
            assert (again == blob).all(), f"request {it}: two runs differ"
            stats["determinism_checks"] += 1
        hb.free()
        stats["requests"] += 1
        stats["new_authorities"] += n_new
        stats["header_bytes"] += e.size
        if it % 10 == 9:
            print(f"soak rotate: {it + 1} / {N} requests, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    stats["seconds"] = round(time.time() - t0, 1)
    print(json.dumps(stats))
    sys.exit(0)
stats = dict(requests=0, verified=0, mutations_refused=0, determinism_checks=0, headers=0, compressions=0, segments={}, profiles={}, modes={})
for it in range(N):
    profile = ["P15k", "Pmix", "Ptiny", "Pmix"][it % 4]
    n_headers = int(rng.integers(1, 257)) if profile != "P15k" else int(rng.choice([16, 64, 256]))
    max_headers = 256
    mode = it % 5  # where the block numbers sit: 1-byte, 2-byte, 4-byte compact, 5-byte, or straddling a boundary
    trusted = [int(rng.integers(1, 40)), int(rng.integers(100, 16000)), int(rng.integers(20000, 1 << 29)), int(rng.integers(1 << 30, (1 << 32) - 600)),
               [63, 16383, (1 << 30) - 1][it % 3] - int(rng.integers(0, max(1, min(n_headers, 30))))][mode]
    trusted = max(1, trusted)
    stride = 35840 if profile != "Ptiny" else 512
    ch = vx.synth.Chain(n_headers, profile=profile, trusted_block=trusted, seed=vx.synth.CHAIN_SEED + 1000 + it, stride=stride)
    n_auth = int(rng.choice([4, 9, 30, 100, 300]))
    need = 2 * n_auth // 3 + 1
    n_signed = int(rng.integers(need, n_auth + 1))
    sj = vx.synth.Justification(ch.target_block, ch.target_hash, n_auth=n_auth, n_signed=n_signed, set_id=int(rng.integers(0, 1 << 40)), seed=vx.synth.JUST_SEED + it)
    just = vx.lib.PackedJustification(sj, 300)
    comps = sum((int(z) + 127) // 128 for z in ch.sizes)
    max_seg = max(1, min(8, n_headers, comps // 4096 if comps >= 8192 else 1))
    n_seg = int(rng.integers(1, max_seg + 1))
    hb = ctx.from_host(ch.headers)
    args = (hb, ch.stride, ch.sizes, max_headers, ch.trusted_block, ch.trusted_hash, ch.target_block, cfg)
    out96, blob = ctx.header_range_prove(*args, just=just, n_segments=n_seg)
    blob = blob.copy()
    assert out96 == ch.expected_outputs(max_headers), f"request {it}: outputs differ from the mirror"
    ver = dict(authority_set_hash=sj.authority_set_hash, authority_set_id=sj.set_id)
    vx.lib.header_range_verify(blob, max_headers, ch.trusted_block, ch.trusted_hash, ch.target_block, out96, cfg, **ver)
    stats["verified"] += 1
    bad = blob.copy()
    bad[int(rng.integers(30, bad.size))] ^= np.uint64(1) << np.uint64(rng.integers(0, 64))
    try:
        vx.lib.header_range_verify(bad, max_headers, ch.trusted_block, ch.trusted_hash, ch.target_block, out96, cfg, **ver)
        raise SystemExit(f"request {it}: a mutated blob was ACCEPTED")
    except vx.VxError:
        stats["mutations_refused"] += 1
    if it % 4 == 0:
        _, again = ctx.header_range_prove(*args, just=just, n_segments=n_seg)
        assert (again == blob).all(), f"request {it}: two runs differ"
        stats["determinism_checks"] += 1
    hb.free()
    stats["requests"] += 1
    stats["headers"] += n_headers
    stats["compressions"] += comps
    for k, v in (("segments", n_seg), ("profiles", profile), ("modes", mode)):
        stats[k][str(v)] = stats[k].get(str(v), 0) + 1
    if it % 10 == 9:
        print(f"soak: {it + 1} / {N} requests, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
stats["seconds"] = round(time.time() - t0, 1)
print(json.dumps(stats))
