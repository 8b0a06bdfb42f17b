#!/bin/bash
# proofs in flight per GPU (bench.py --inflight): throughput and the latency of a proof alone
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for k in 2 3 4 5 6 8; do
  python3 $R/bench.py --steps 24 --warmup 6 --no-cpu-baseline --no-pmax --inflight $k > $O/ab_inflight_$k.json 2>/dev/null
  python3 -c "import json; d=json.loads([l for l in open('$O/ab_inflight_$k.json') if l.startswith('{')][-1]); print('inflight $k:', d['value'], 'proofs/s', d['ms_per_step'], 'ms per step')"
done | tee $O/r03_ab_inflight.txt
