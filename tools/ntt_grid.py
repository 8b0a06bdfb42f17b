#!/usr/bin/env python3
"""NTT microbenchmark grid of SURVEY.md section 8d: log2 n in {12,...,24} x columns {1,16,80,135,256}, elements uniform
in [0,p) (seed 42), forward / inverse / LDE (rate_bits 1 = this prover's config, 3 = plonky2's standard config, shift 7).
Algorithmic bytes: NTT 16*n*c, LDE 8*n*c*(1+2^r); time from HIP events on the ctx stream.  Writes JSON to stdout."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import vx_import  # noqa: E402

vx = vx_import.load()
ctx = vx.Context(0)
rows = []
for log_n in (12, 14, 16, 18, 20, 22, 24):
    for c in (1, 16, 80, 135, 256):
        n = 1 << log_n
        src = ctx.alloc(n * c)
        ctx.fill_random(src, n * c, 42)
        iters = max(2, min(50, int(2e9 / (n * c * 16))))

        def timed(f):
            f()
            ctx.sync()
            ctx.timer_start()
            for _ in range(iters):
                f()
            return ctx.timer_stop() / iters

        rec = {"log_n": log_n, "cols": c}
        t = timed(lambda: ctx.ntt(src, log_n, c, order=1))
        rec["fwd_ms"], rec["fwd_GBs"] = round(t, 4), round(16.0 * n * c / t / 1e6, 1)
        t = timed(lambda: ctx.ntt(src, log_n, c, inverse=True, order=1))
        rec["inv_ms"], rec["inv_GBs"] = round(t, 4), round(16.0 * n * c / t / 1e6, 1)
        for r in (1, 3):
            if (n * c << r) * 8 > 96e9:
                continue
            dst = ctx.alloc((n * c) << r)
            t = timed(lambda: ctx.lde(src, log_n, c, r, dst))
            rec[f"lde_r{r}_ms"], rec[f"lde_r{r}_GBs"] = round(t, 4), round(8.0 * n * c * (1 + (1 << r)) / t / 1e6, 1)
            dst.free()
        src.free()
        rows.append(rec)
        print(json.dumps(rec), file=sys.stderr, flush=True)
print(json.dumps({"unit": "GB/s algorithmic (SURVEY 8d byte counts); ms per transform", "peak_GBs": 8000, "grid": rows}))
