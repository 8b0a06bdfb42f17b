#!/usr/bin/env python3
"""A/B timing of the hot kernels on one shape (HIP events on the ctx stream):
Poseidon leaf hashing (2^18 leaves x 4337 columns), LDE 2^17 x 1024 -> 2^18, forward NTT 2^19 x 1024.
usage: python3 tools/ab_kernels.py [tag]   (VX_LIB_PATH=... selects another libvxprove build)"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import vx_import  # noqa: E402

vx = vx_import.load()
ctx = vx.Context(0)
tag = sys.argv[1] if len(sys.argv) > 1 else "a"
res = {"tag": tag, "lib": vx.lib.LIB_PATH}


def timed(f, iters=5):
    f()
    ctx.sync()
    ctx.timer_start()
    for _ in range(iters):
        f()
    return ctx.timer_stop() / iters


L, c = 18, 4337
buf = ctx.alloc((1 << L) * c)
ctx.fill_random(buf, (1 << L) * c, 11)
trees = []


def hash_leaves():
    t = ctx.merkle(buf, 1 << L, c, 1, 4)
    trees.append(t)


res["merkle_2^18x4337_ms"] = round(timed(hash_leaves, 3), 3)
for t in trees:
    t.free()
buf.free()
n, c2 = 19, 1024
b2 = ctx.alloc((1 << n) * c2)
ctx.fill_random(b2, (1 << n) * c2, 5)
res["ntt_fwd_2^19x1024_ms"] = round(timed(lambda: ctx.ntt(b2, n, c2, order=1)), 3)
res["ntt_inv_2^19x1024_ms"] = round(timed(lambda: ctx.ntt(b2, n, c2, inverse=True, order=1)), 3)
b3 = ctx.alloc((1 << (n + 1)) * c2)
res["lde_2^19x1024_r1_ms"] = round(timed(lambda: ctx.lde(b2, n, c2, 1, b3)), 3)
print(json.dumps(res))
