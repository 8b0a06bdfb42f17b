#!/bin/bash
# End-of-round batch (through gpurun, from the repo root): the NTT roofline shape under rocprofv3 (kernel_stats.csv of the FINAL build),
# smoke(), the default bench command with its wall time, the other bench configurations, one proof's kernel stats, the NTT / LDE grid.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R && python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r03_smoke.log 2>&1 || { tail -5 $O/r03_smoke.log; exit 1; }
bash $R/tools/r03_ntt_probe.sh final > $O/r03_ntt_probe_final.txt 2>&1
cp $(find $O/r03_ntt_final -name "*kernel_stats.csv" | head -1) $O/r03_ntt_kernel_stats_final.csv
cd /tmp && export TMPDIR=/tmp
s=$(date +%s%N); python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/r03_driver_cmd_bench.json 2> $O/r03_driver_cmd_bench.err; e=$(date +%s%N)
echo "wall of the driver command: $(( (e - s) / 1000000 )) ms" > $O/r03_driver_cmd_wall.txt
bash $R/tools/r03_profile_batch.sh > $O/r03_profile_batch.log 2>&1
python3 $R/tools/ntt_grid.py > $O/r03_ntt_grid.json 2> $O/r03_ntt_grid.err
cat $O/r03_driver_cmd_wall.txt; tail -2 $O/r03_ntt_probe_final.txt | head -1; head -c 300 $O/r03_final_bench.json; echo; echo batch done
