#!/bin/bash
# rocprofv3 --kernel-trace --stats of the NTT roofline shape alone (tools/ntt_stats_run.py: 32 forward NTTs of 2^19 x 1024 = 64 launches of
# k_ntt_tile) and of the default bench command: the rocprofv3-native kernel_stats.csv files copied to profiles/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_ntt_stats -o p -- python3 $R/tools/ntt_stats_run.py > $O/r02_ntt_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_bench_stats -o p -- python3 $R/bench.py --no-cpu-baseline > $O/r02_bench_stats.json 2> $O/r02_bench_stats.log
find $O/r02_ntt_stats $O/r02_bench_stats -name "*stats*.csv" | head
echo done
