#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for q in 256 512 1024; do
  if [ $q = 256 ]; then unset VX_LIB_PATH; else export VX_LIB_PATH=$R/0-kno-vectorx_amd/libvx_q$q.so; fi
  rocprofv3 --kernel-trace --stats -d $O/abq_$q -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 > $O/abq_$q.log 2>&1
  python3 $R/tools/rocpd_timeline.py $O/abq_$q | grep -E "one proof|k_quotient" > $O/abq_$q.txt
  echo "== $q"; cat $O/abq_$q.txt
done
