#!/usr/bin/env python3
"""The NTT roofline shape of bench.py alone, for `rocprofv3 --kernel-trace --stats`: 32 forward NTTs of 2^19 x 1024 (64 launches of
k_ntt_tile<0, 0>), so that the profiler's average launch duration can be set beside bench.py's HIP-event `roofline.ms_per_launch`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vx_import  # noqa: E402

vx = vx_import.load()
with vx.Context(0) as ctx:
    n = (1 << 19) * 1024
    a = ctx.alloc(n)
    ctx.fill_random(a, n, 7)
    for _ in range(32):
        ctx.ntt(a, 19, 1024, order=1)
    ctx.sync()
