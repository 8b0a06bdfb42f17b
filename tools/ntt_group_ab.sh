#!/bin/bash
# A/B of VX_NTT_GROUP (all passes of a transform over G columns before the next G): time from bench.py's NTT roofline probe,
# HBM traffic from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/ntt_pmc.py (separate passes, see profiles/README.md)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for g in 0 16 32 64; do
  export VX_NTT_GROUP=$g
  python3 - <<PY > $O/nttg_$g.time 2>&1
import sys, time
sys.path.insert(0, "$R")
import vx_import
vx = vx_import.load()
with vx.Context(0) as ctx:
    n = (1 << 19) * 1024
    a = ctx.alloc(n); ctx.fill_random(a, n, 1)
    for _ in range(3): ctx.ntt(a, 19, 1024, order=1)
    ctx.sync(); t = time.perf_counter()
    for _ in range(10): ctx.ntt(a, 19, 1024, order=1)
    ctx.sync(); dt = (time.perf_counter() - t) / 10
    print("group", $g, "ms_per_transform", round(1e3 * dt, 3), "GB/s algorithmic", round(16 * n / dt / 1e9, 1))
PY
  cat $O/nttg_$g.time
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/nttg_${g}_fetch -o p -- python3 $R/tools/ntt_pmc.py > $O/nttg_${g}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/nttg_${g}_write -o p -- python3 $R/tools/ntt_pmc.py > $O/nttg_${g}_write.log 2>&1
done
echo done
