#!/usr/bin/env python3
"""Wall time of header_range_256 proofs for A/B builds (VX_LIB_PATH); prints ms per proof."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import vx_import
vx = vx_import.load()
ch = vx.synth.Chain(256, profile="P15k")
ctx = vx.Context(0)
hb = ctx.from_host(ch.headers)
cfg = ctx.stark_config()
out = None
for i in range(4):
    if i == 1:
        ctx.sync(); t = time.perf_counter()
    o, blob = ctx.header_range_prove(hb, ch.stride, ch.sizes, 256, ch.trusted_block, ch.trusted_hash, ch.target_block, cfg, out)
    out = blob.base if blob.base is not None else blob
ctx.sync()
print(os.path.basename(vx.lib.LIB_PATH), round((time.perf_counter() - t) / 3 * 1e3, 2), "ms per proof (no justification)")
