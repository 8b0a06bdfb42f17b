#!/bin/bash
# runs tools/ntt_ab.py for the default library and every build_ab/libvx_*.so; one JSON line each
R=$GRAFT_REPO_ROOT
python3 $R/tools/ntt_ab.py default
for f in $R/build_ab/libvx_*.so; do t=$(basename $f .so); VX_LIB_PATH=$f python3 $R/tools/ntt_ab.py ${t#libvx_} || exit 1; done
