#!/bin/bash
# Round-2 measurement batch (run through gpurun from the repo root): bench lines + rocprofv3 kernel stats + NTT PMC passes.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/r02_final_bench.json 2> $O/r02_final_bench.err
python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --inflight 1 > $O/r02_bench_inflight1.json 2>> $O/r02_final_bench.err
python3 $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --headers 512 > $O/r02_bench_header_range_512.json 2>> $O/r02_final_bench.err
python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --circuit rotate > $O/r02_bench_rotate.json 2>> $O/r02_final_bench.err
VX_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29513 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 python3 $R/bench.py --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline > $O/r02_bench_rccl_path_1gpu.json 2>> $O/r02_final_bench.err
rocprofv3 --kernel-trace --stats -d $O/r02_prof_final -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 > $O/r02_prof_final.log 2>&1
VX_BENCH_NO_JUSTIFICATION=1 rocprofv3 --kernel-trace --stats -d $O/r02_prof_hashchain -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 > $O/r02_prof_hashchain.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r02_pmc_fetch -o p -- python3 $R/tools/ntt_pmc.py > $O/r02_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r02_pmc_write -o p -- python3 $R/tools/ntt_pmc.py > $O/r02_pmc_write.log 2>&1
echo batch done
