#!/usr/bin/env python3
"""A/B timing of the NTT shapes a header_range proof uses (HIP events on the ctx stream): forward / inverse 2^19 x 1024,
LDE 2^19 -> 2^20 x 1024.  usage: python3 tools/ntt_ab.py [tag]   (VX_LIB_PATH=... selects another libvxprove build,
tools/ntt_ab_build.sh makes them from -D variants of vx_ntt.hip)"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import vx_import  # noqa: E402

vx = vx_import.load()
ctx = vx.Context(0)
res = {"tag": sys.argv[1] if len(sys.argv) > 1 else "a", "lib": os.path.basename(vx.lib.LIB_PATH)}


def timed(f, iters=10):
    for _ in range(2):
        f()
    ctx.sync()
    best = 1e9
    for _ in range(3):
        ctx.timer_start()
        for _ in range(iters):
            f()
        best = min(best, ctx.timer_stop() / iters)
    return round(best, 3)


n, c = 19, 1024
b = ctx.alloc((1 << n) * c)
ctx.fill_random(b, (1 << n) * c, 5)
res["fwd"] = timed(lambda: ctx.ntt(b, n, c, order=1))
res["inv"] = timed(lambda: ctx.ntt(b, n, c, inverse=True, order=1))
b3 = ctx.alloc((1 << (n + 1)) * c)
res["lde_r1"] = timed(lambda: ctx.lde(b, n, c, 1, b3), 5)
res["fwd_frac_of_8TBs"] = round(16 * (1 << n) * c / (res["fwd"] * 1e-3) / 8e12, 4)
print(json.dumps(res))
