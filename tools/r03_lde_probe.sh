#!/bin/bash
# per-kernel average durations of the LDE shape (tools/lde_stats_run.py) from a rocprofv3 kernel trace.  $1 = tag, VX_LIB_PATH honoured
TAG=${1:-base}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_lde_$TAG -o p -- python3 $R/tools/lde_stats_run.py > $O/r03_lde_$TAG.log 2>&1 || exit 1
python3 - "$O/r03_lde_$TAG" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
acc = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"][:48]
    acc.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
tot = 0
for k, v in acc.items():
    v2 = v[len(v) // 4:]
    print("%-50s n %3d avg ms %.3f" % (k, len(v), sum(v2) / len(v2)))
    if "ntt" in k: tot += sum(v2) / len(v2) * (len(v) / 8)
print("ntt kernels per LDE: %.3f ms" % tot)
PY
