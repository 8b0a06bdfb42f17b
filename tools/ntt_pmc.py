#!/usr/bin/env python3
"""NTT traffic measurement helper (run under rocprofv3 --pmc ...; see profiles/README.md).

Launches, in order: a CALIBRATION pair with known byte counts in the same 8-B-per-lane access
pattern as the NTT (k_fill_random: writes 8*n*c bytes; k_batch<0> field add: reads 16*n*c, writes
8*n*c), then 3 forward NTTs (bit-reversed output, 2 launches of k_ntt_tile each) of 2^19 x 1024.
The gfx950 FETCH_SIZE / WRITE_SIZE counters are calibrated on the pair (MI355X_MICROARCH.md, HBM
section: FETCH_SIZE is uncalibrated for widths other than 16 B/lane) by tools/ntt_pmc_report.py.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vx_import  # noqa: E402

LOG_N, COLS = 19, 1024


def main():
    vx = vx_import.load()
    with vx.Context(0) as ctx:
        n = (1 << LOG_N) * COLS
        a, b, c = ctx.alloc(n), ctx.alloc(n), ctx.alloc(n)
        ctx.fill_random(a, n, 1)
        ctx.fill_random(b, n, 2)
        ctx.field_op("add", a, b, c, n)
        for _ in range(3):
            ctx.ntt(a, LOG_N, COLS, order=1)
        ctx.sync()


if __name__ == "__main__":
    main()
