#!/bin/bash
# A/B of k_quotient's register budget (VX_Q_WAVES = 4 default / 3 / 2 waves per SIMD): libvx_qw{2,3}.so are builds of the library
# whose vx_stark.hip was compiled with -DVX_Q_WAVES=N (tools/README.md).  Prints the quotient kernels and the wall of one proof.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for w in 4 3 2; do
  if [ $w = 4 ]; then unset VX_LIB_PATH; else export VX_LIB_PATH=$R/0-kno-vectorx_amd/libvx_qw$w.so; fi
  rocprofv3 --kernel-trace --stats -d $O/abqw_$w -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 > $O/abqw_$w.log 2>&1
  python3 $R/tools/rocpd_timeline.py $O/abqw_$w | grep -E "one proof|k_quotient" > $O/abqw_$w.txt
  echo "== waves $w"; cat $O/abqw_$w.txt
  python3 $R/bench.py --steps 9 --warmup 3 --no-cpu-baseline > $O/abqw_${w}_bench.json 2>/dev/null
  python3 -c "import json,sys; d=json.loads([l for l in open('$O/abqw_${w}_bench.json') if l.startswith('{')][-1]); print('throughput', d['value'], 'latency', d['latency_ms'])"
done
