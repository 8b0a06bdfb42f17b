#!/usr/bin/env python3
"""Timing experiment: forward NTT 2^19 x 1024 (DIF, bit-reversed output) with parts of the kernel
skipped through VX_NTT_SKIP (results are WRONG when skipping; timing only)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys; sys.path.insert(0, %r)
import vx_import
vx = vx_import.load()
with vx.Context(0) as ctx:
    n, c = 19, 1024
    buf = ctx.alloc(c << n); ctx.fill_random(buf, c << n, 3)
    for _ in range(2): ctx.ntt(buf, n, c, order=1)
    ctx.sync(); ctx.timer_start()
    for _ in range(10): ctx.ntt(buf, n, c, order=1)
    print("%%.3f" %% (ctx.timer_stop() / 10))
''' % ROOT
for skip, name in ((0, "full"), (1, "no tile twiddles"), (2, "no inter-pass twiddle"), (4, "no butterflies"), (7, "memory + exchanges only")):
    out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, VX_NTT_SKIP=str(skip)), capture_output=True, text=True)
    print(f"{name:28s} {out.stdout.strip()} ms  {out.stderr[-200:] if out.returncode else ''}")
