#!/bin/bash
# Round-3 NTT probe: per-LAUNCH durations of the roofline shape (pass 1 = strided 7-stage tile pass, pass 2 = contiguous 12-stage pass)
# from a rocprofv3 kernel trace, and the VX_NTT_SKIP breakdown.  $1 = tag for the output files.
TAG=${1:-base}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_ntt_$TAG -o p -- python3 $R/tools/ntt_stats_run.py > $O/r03_ntt_$TAG.log 2>&1 || exit 1
python3 - "$O/r03_ntt_$TAG" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_ntt" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
ev, od = dur[4::2], dur[5::2]  # skip the first two transforms
print("launches", len(dur), "pass1 avg ms %.3f  pass2 avg ms %.3f  sum %.3f" % (sum(ev) / len(ev), sum(od) / len(od), sum(ev) / len(ev) + sum(od) / len(od)))
print("names", collections.Counter(r["Kernel_Name"][:60] for r in rows))
print("vgpr/sgpr/lds", {(r["Kernel_Name"][:40], r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size")) for r in rows})
PY
python3 $R/tools/ntt_breakdown.py
echo done
