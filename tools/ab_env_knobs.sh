#!/bin/bash
# ROCm runtime knobs that touch launch latency: kernel arguments in device memory (HIP_FORCE_DEV_KERNARG), SDMA engines for copies.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() {
  python3 $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline > $O/abenv_$1.json 2>/dev/null
  python3 -c "import json; d=json.loads([l for l in open('$O/abenv_$1.json') if l.startswith('{')][-1]); print('$1: throughput', d['value'], 'latency', d['latency_ms'])"
}
run default
HIP_FORCE_DEV_KERNARG=1 run dev_kernarg1
HIP_FORCE_DEV_KERNARG=0 run dev_kernarg0
HSA_ENABLE_SDMA=0 run sdma0
run default_again
