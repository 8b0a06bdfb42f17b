R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU --output-format csv -d $O/r03_pmc_valu/a -o p -- python3 $R/tools/prove_once.py 256 1 > $O/r03_pmc_valu_a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/r03_pmc_valu/b -o p -- python3 $R/tools/prove_once.py 256 1 > $O/r03_pmc_valu_b.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $O/r03_pmc_valu/c -o p -- python3 $R/tools/prove_once.py 256 1 > $O/r03_pmc_valu_c.log 2>&1
echo pmc batch done
python3 $R/tools/pmc_kernel_report.py $O/r03_pmc_valu $O/r03_pmc_valu_by_kernel.json > $O/r03_pmc_valu_report.txt 2>&1
