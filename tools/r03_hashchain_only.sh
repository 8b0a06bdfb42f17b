#!/bin/bash
# Kernel trace of ONE proof with the hash-chain table alone (VX_BENCH_NO_JUSTIFICATION=1: a profiling aid, not the metric):
# single stream, so the durations are the kernels' own.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
VX_BENCH_NO_JUSTIFICATION=1 rocprofv3 --kernel-trace --stats -d $O/r03_prof_hc -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmax --inflight 1 > $O/r03_prof_hc.log 2>&1
python3 $R/tools/rocpd_timeline.py $O/r03_prof_hc > $O/r03_hashchain_only_kernel_stats.txt 2>&1
echo done
