#!/usr/bin/env python3
"""One header_range proof (statement checks + both STARKs) for rocprofv3 --pmc passes.
usage: python3 tools/prove_once.py [n_headers] [n_proofs]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import vx_import  # noqa: E402

vx = vx_import.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ch = vx.synth.Chain(n, profile="P15k")
ctx = vx.Context(0)
hb = ctx.from_host(ch.headers)
cfg = ctx.stark_config()
just = vx.lib.PackedJustification(vx.synth.Justification(ch.target_block, ch.target_hash, n_auth=300), 300)
for _ in range(reps):
    out96, blob = ctx.header_range_prove(hb, ch.stride, ch.sizes, n, ch.trusted_block, ch.trusted_hash, ch.target_block, cfg, just=just)
assert out96 == ch.expected_outputs(n)
print("ok", blob.size)
