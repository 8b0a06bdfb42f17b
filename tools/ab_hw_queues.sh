#!/bin/bash
# Does the HIP runtime's hardware-queue limit (GPU_MAX_HW_QUEUES, default 4) cap the overlap of the 15 streams of three proofs in flight?
# usage: ab_hw_queues.sh "<queue counts>" "<inflight counts>"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for q in ${1:-default 2 8 16}; do
  for inf in ${2:-3 4}; do
    if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --inflight $inf > $O/abhq_${q}_$inf.json 2>/dev/null
    python3 -c "import json; d=json.loads([l for l in open('$O/abhq_${q}_$inf.json') if l.startswith('{')][-1]); print('queues $q inflight $inf: throughput', d['value'], 'latency', d['latency_ms'])"
  done
done
