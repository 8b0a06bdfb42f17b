#!/bin/bash
# Round-3 PMC batch for the NTT roofline shape (forward 2^19 x 1024): HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes,
# calibrated, tools/ntt_pmc.py) and SQ issue counters, for the round-3 kernel (k_ntt3) and, with VX_NTT_V2=1, the round-2
# run-time-shape kernel (k_ntt_tile) it replaces.  Outputs under gpurun_out/r03_pmc_*; reports by tools/ntt_pmc_report.py
# and tools/pmc_kernel_report.py.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for V in ${@:-new old}; do
  if [ $V = old ]; then export VX_NTT_V2=1; else unset VX_NTT_V2; fi
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r03_pmc_$V/fetch -o p -- python3 $R/tools/ntt_pmc.py > $O/r03_pmc_${V}_f.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r03_pmc_$V/write -o p -- python3 $R/tools/ntt_pmc.py > $O/r03_pmc_${V}_w.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU --output-format csv -d $O/r03_pmc_$V/sq/a -o p -- python3 $R/tools/ntt_pmc.py > $O/r03_pmc_${V}_a.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/r03_pmc_$V/sq/b -o p -- python3 $R/tools/ntt_pmc.py > $O/r03_pmc_${V}_b.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU --output-format csv -d $O/r03_pmc_$V/sq/c -o p -- python3 $R/tools/ntt_pmc.py > $O/r03_pmc_${V}_c.log 2>&1 || exit 1
  python3 $R/tools/ntt_pmc_report.py $O/r03_pmc_$V/fetch $O/r03_pmc_$V/write $O/r03_ntt_traffic_$V.json
  python3 $R/tools/pmc_kernel_report.py $O/r03_pmc_$V/sq $O/r03_ntt_sq_$V.json
done
echo pmc done
