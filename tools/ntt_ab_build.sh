#!/bin/bash
# Builds A/B variants of libvxprove.so that differ only in vx_ntt.o:  tools/ntt_ab_build.sh tag "-DFLAG=.." [tag2 "-D.." ...]
# -> build_ab/libvx_<tag>.so (travels to the GPU box; select with VX_LIB_PATH).  Run after `make` so the other objects exist.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/0-kno-vectorx_amd/csrc; mkdir -p $R/build_ab
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-pass-failed -I$R/include $flags -c $C/vx_ntt.hip -o $R/build_ab/vx_ntt_$tag.o 2>&1 | grep -v hip-link || true
    objs=$(ls $C/*.o | grep -v vx_ntt.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_ab/libvx_$tag.so $objs $R/build_ab/vx_ntt_$tag.o 2>&1 | grep -v hip-link || true; echo built $tag ) &
done
wait
