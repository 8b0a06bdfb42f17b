#!/usr/bin/env python3
"""Kernel timeline / per-kernel totals of ONE proof from a rocprofv3 (rocpd sqlite) kernel trace.
usage: rocpd_timeline.py <results.db> [--timeline]   (the last proof in the trace: from its last k_blake2b_256 on)"""
import glob
import sqlite3
import sys

db = sys.argv[1]
if not db.endswith(".db"):
    db = glob.glob(db + "/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
rows = con.execute("select name, start, end, grid_x, workgroup_x, vgpr_count, scratch_size from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if r[0].startswith("k_blake2b_256")]
s = idx[-1] if idx else 0
t0 = rows[s][1]
# the proof ends where bench.py's micro-benchmarks begin (they fill their buffers with k_fill_random)
end = next((i for i in range(s, len(rows)) if rows[i][0].startswith("k_fill_random")), len(rows))
rows = rows[:end]
agg = {}
for r in rows[s:]:
    nm = r[0].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:44]
    d = (r[2] - r[1]) / 1e3
    if "--timeline" in sys.argv:
        print(f"{(r[1] - t0) / 1e6:9.3f} ms  {d:9.1f} us  {nm:46s} grid={r[3]:>10} vgpr={r[5]} scratch={r[6]}")
    a = agg.setdefault(nm, [0, 0.0])
    a[0] += 1
    a[1] += d
tot = sum(a[1] for a in agg.values())
span = (rows[-1][2] - t0) / 1e6
print(f"--- one proof: {span:.1f} ms wall on the stream, {tot / 1e3:.1f} ms of kernels")
for nm, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{d / 1e3:9.2f} ms {100 * d / tot:5.1f}%  x{n:<4d} {nm}")
