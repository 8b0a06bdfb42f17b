#!/bin/bash
# Is the library constructor early enough?  (a) a C-like host: no Python setdefault (VX_NO_PY_ENV=1 makes lib.py skip it); (b) the default path
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
unset GPU_MAX_HW_QUEUES
VX_NO_PY_ENV=1 python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline > $O/abhq_ctor.json 2>/dev/null
python3 -c "import json; d=json.loads([l for l in open('$O/abhq_ctor.json') if l.startswith('{')][-1]); print('constructor only: throughput', d['value'], 'latency', d['latency_ms'], 'inflight', d['inflight_per_gpu'])"
python3 $R/bench.py > $O/abhq_defaultrun.json 2>/dev/null
python3 -c "import json; d=json.loads([l for l in open('$O/abhq_defaultrun.json') if l.startswith('{')][-1]); print('default run: throughput', d['value'], 'latency', d['latency_ms'], 'inflight', d['inflight_per_gpu'], 'steps', d['steps'])"
GPU_MAX_HW_QUEUES=4 python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline > $O/abhq_4.json 2>/dev/null
python3 -c "import json; d=json.loads([l for l in open('$O/abhq_4.json') if l.startswith('{')][-1]); print('host chose 4: throughput', d['value'], 'latency', d['latency_ms'])"
