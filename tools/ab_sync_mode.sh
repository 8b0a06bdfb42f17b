#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
nproc
for m in default b y s; do
  if [ $m = default ]; then unset VX_SYNC_MODE; else export VX_SYNC_MODE=$m; fi
  time python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline > $O/absync_$m.json 2> $O/absync_$m.err
  python3 -c "import json; d=json.loads([l for l in open('$O/absync_$m.json') if l.startswith('{')][-1]); print('sync $m: throughput', d['value'], 'latency', d['latency_ms'])"
  tail -2 $O/absync_$m.err
done
