#!/bin/bash
# kernel trace of the throughput configuration (four proofs in flight) -> how busy the GPU is (tools/rocpd_busy.py)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/r03_prof_busy -o p -- python3 $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-pmax > $O/r03_prof_busy.log 2>&1
python3 $R/tools/rocpd_busy.py $O/r03_prof_busy > $O/r03_gpu_busy.txt 2>&1
rm -rf $O/r03_prof_busy
cat $O/r03_gpu_busy.txt
