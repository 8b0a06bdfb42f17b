#!/bin/bash
# Round-3 measurement batch (run through gpurun from the repo root): bench lines of every configuration + rocprofv3 kernel stats of one proof.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/r03_final_bench.json 2> $O/r03_final_bench.err
python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pmax --inflight 1 > $O/r03_bench_inflight1.json 2>> $O/r03_final_bench.err
python3 $R/bench.py --steps 6 --warmup 1 --headers 512 > $O/r03_bench_header_range_512.json 2>> $O/r03_final_bench.err
python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --profile Pmax > $O/r03_bench_pmax.json 2>> $O/r03_final_bench.err
python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --circuit rotate > $O/r03_bench_rotate.json 2>> $O/r03_final_bench.err
VX_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29513 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 python3 $R/bench.py --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline --no-pmax > $O/r03_bench_rccl_path_1gpu.json 2>> $O/r03_final_bench.err
rocprofv3 --kernel-trace --stats -d $O/r03_prof_final -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmax --inflight 1 > $O/r03_prof_final.log 2>&1
python3 $R/tools/rocpd_timeline.py $O/r03_prof_final > $O/r03_final_kernel_stats.txt 2>&1
echo batch done
