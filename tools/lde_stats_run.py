#!/usr/bin/env python3
"""The LDE shape of a header_range_256 proof alone, for `rocprofv3 --kernel-trace --stats`: 8 x (values 2^19 x 1024 -> coset
evaluations 2^20 x 1024, rate_bits 1) = inverse DIF (2 launches) + forward DIT with zero padding and coset scaling (2 launches)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vx_import  # noqa: E402

vx = vx_import.load()
with vx.Context(0) as ctx:
    n, c = 19, 1024
    a = ctx.alloc(c << n)
    b = ctx.alloc(c << (n + 1))
    ctx.fill_random(a, c << n, 7)
    for _ in range(8):
        ctx.lde(a, n, c, 1, b)
    ctx.sync()
