#!/bin/bash
# A/B of k_hash_leaves' register budget (VX_HASH_WAVES = 6 default / 4 / 5 / 8 waves per SIMD): libvx_hw{N}.so = the library with vx_poseidon.hip
# compiled with -DVX_HASH_WAVES=N, selected with VX_LIB_PATH.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for w in 6 4 5 8; do
  if [ $w = 6 ]; then unset VX_LIB_PATH; else export VX_LIB_PATH=$R/0-kno-vectorx_amd/libvx_hw$w.so; fi
  python3 $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline > $O/abhw_$w.json 2>/dev/null
  python3 -c "import json; d=json.loads([l for l in open('$O/abhw_$w.json') if l.startswith('{')][-1]); print('hash waves $w: throughput', d['value'], 'latency', d['latency_ms'], 'poseidon', d['roofline_poseidon']['achieved'], d['roofline_poseidon']['per'][-9:])"
done
