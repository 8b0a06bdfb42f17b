#!/usr/bin/env python3
"""GPU occupancy of a throughput run from a rocprofv3 (rocpd sqlite) kernel trace: over the window of the timed proofs, the
fraction of wall time with at least one kernel running, the histogram of how many run at once, and which kernels run ALONE.
usage: rocpd_busy.py <results.db or directory> [skip_first_fraction=0.3]"""
import glob
import sqlite3
import sys

db = sys.argv[1]
if not db.endswith(".db"):
    db = glob.glob(db + "/**/*.db", recursive=True)[0]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows = sqlite3.connect(db).execute("select name, start, end from kernels order by start").fetchall()
end_i = next((i for i, r in enumerate(rows) if r[0].startswith("k_fill_random")), len(rows))  # bench.py's micro-benchmarks follow
rows = rows[:end_i]
t0, t1 = rows[0][1], max(r[2] for r in rows)
lo = t0 + skip * (t1 - t0)  # warm-up and start-up are not steady state
ev = []
for nm, s, e in rows:
    if e <= lo:
        continue
    ev.append((max(s, lo), 1, nm)), ev.append((e, -1, nm))
ev.sort(key=lambda x: (x[0], x[1]))
hist, alone, cur, last, running = {}, {}, 0, lo, {}
for t, d, nm in ev:
    dt = t - last
    if dt > 0:
        hist[cur] = hist.get(cur, 0) + dt
        if cur == 1:
            k = next(iter(running))
            alone[k] = alone.get(k, 0) + dt
    last = t
    cur += d
    if d > 0:
        running[nm] = running.get(nm, 0) + 1
    else:
        running[nm] -= 1
        if running[nm] == 0:
            del running[nm]
wall = t1 - lo
print(f"window {wall / 1e6:.1f} ms; idle {100 * hist.get(0, 0) / wall:.1f} %; kernels at once (share of wall): " +
      ", ".join(f"{k}: {100 * v / wall:.1f} %" for k, v in sorted(hist.items())))
short = lambda n: n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:40]
print("alone on the GPU (ms): " + ", ".join(f"{short(k)} {v / 1e6:.1f}" for k, v in sorted(alone.items(), key=lambda kv: -kv[1])[:8]))
