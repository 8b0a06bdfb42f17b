#!/bin/bash
# A/B of the partial rounds of the Poseidon permutation: eleven passive elements recombined to 64-bit words after every layer (0) vs kept
# split (low part + high part 2^52) and only normalised between layers (VX_POSEIDON_SPLIT_PARTIAL=1).  `build` (no GPU needed) makes
# build_ab/poseidon_rate_sp{0,1} and libvx_psp{0,1}.so; `run` (through gpurun) prints the microbenchmark lines and bench.py with each library.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; C=$R/0-kno-vectorx_amd/csrc
if [ "$1" = build ]; then
  mkdir -p $R/build_ab
  for v in 0 1; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVX_POSEIDON_SPLIT_PARTIAL=$v -I$R/include -o $R/build_ab/poseidon_rate_sp$v $R/tools/poseidon_rate.hip
    for f in vx_poseidon vx_fri; do
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-value -Wno-pass-failed -DVX_POSEIDON_SPLIT_PARTIAL=$v -I$R/include -c $C/$f.hip -o $R/build_ab/${f}_psp$v.o
    done
    objs=""
    for o in $C/*.o; do b=$(basename $o .o); if [ -f $R/build_ab/${b}_psp$v.o ]; then objs="$objs $R/build_ab/${b}_psp$v.o"; else objs="$objs $o"; fi; done
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/0-kno-vectorx_amd/libvx_psp$v.so $objs
  done
  exit 0
fi
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do $R/build_ab/poseidon_rate_sp$v; done | tee $O/ab_poseidon_split_micro.txt
for v in 0 1; do
  export VX_LIB_PATH=$R/0-kno-vectorx_amd/libvx_psp$v.so
  python3 $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-pmax > $O/ab_psp_$v.json 2>/dev/null
  python3 -c "import json; d=json.loads([l for l in open('$O/ab_psp_$v.json') if l.startswith('{')][-1]); print('split=$v: proofs/s', d['value'], 'latency', d['latency_ms'], 'poseidon', d['roofline_poseidon']['achieved'], d['roofline_poseidon']['unit'])"
done | tee $O/ab_poseidon_split_bench.txt
