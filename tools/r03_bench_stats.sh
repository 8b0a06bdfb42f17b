#!/bin/bash
# rocprofv3 --kernel-trace --stats of the DEFAULT bench command (the rocprofv3-native kernel_stats.csv): the rows of k_ntt3<0, 0, 7, 0> and
# k_ntt3<0, 0, 12, 0> are the roofline micro-benchmark inside bench.py -- their averages are what `roofline.ms_per_launch` of the line
# printed by the same run must agree with.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_bench_stats -o p -- python3 $R/bench.py --no-cpu-baseline --no-pmax > $O/r03_bench_under_rocprof.json 2> $O/r03_bench_stats.log
cp $(find $O/r03_bench_stats -name "*kernel_stats.csv" | head -1) $O/r03_bench_kernel_stats.csv
grep "k_ntt3<0, 0" $O/r03_bench_kernel_stats.csv
python3 -c "import json; d=json.loads([l for l in open('$O/r03_bench_under_rocprof.json') if l.startswith('{')][-1]); print(d['value'], d['roofline']['ms_per_transform'], d['roofline']['ms_per_transform_groups'])"
rm -rf $O/r03_bench_stats
