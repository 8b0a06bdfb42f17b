#!/bin/bash
# A/B of stream priorities (VX_STREAM_PRIO=1: the host's context -- the hash-chain table -- highest, the side contexts of the other
# tables lowest): single-proof latency and throughput of bench.py.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for p in 0 1 0 1; do
  export VX_STREAM_PRIO=$p
  python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-pmax > $O/ab_prio_$p.json 2>/dev/null
  python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pmax --inflight 1 > $O/ab_prio_${p}_i1.json 2>/dev/null
  python3 -c "import json; d=json.loads([l for l in open('$O/ab_prio_$p.json') if l.startswith('{')][-1]); e=json.loads([l for l in open('$O/ab_prio_${p}_i1.json') if l.startswith('{')][-1]); print('prio $p: 4 in flight', d['value'], 'proofs/s, latency', d['latency_ms'], 'ms; 1 in flight', e['value'], 'proofs/s', e['ms_per_step'], 'ms')"
done | tee $O/ab_stream_prio.txt
