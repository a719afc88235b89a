#!/bin/bash
# Dev tool: variant library with ONE translation unit recompiled with extra flags, the rest taken from the in-tree build:
#   scripts/dev/ab_one.sh <unit (e.g. lipvq_mlp)> name "<flags>" [name2 "<flags2>" ...]  ->  build_ab/<name>/_lipvq_hip.so
set -e
cd "$(dirname "$0")/../.."
unit=$1; shift
make -s -C lipvq-vae_amd/csrc
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function -fno-slp-vectorize"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  mkdir -p build_ab/$name
  ( /opt/rocm/bin/hipcc $BASE $flags -c -o build_ab/$name/$unit.o lipvq-vae_amd/csrc/$unit.hip
    objs=""
    for o in lipvq-vae_amd/csrc/build/*.o; do [ "$(basename $o .o)" = "$unit" ] && objs="$objs build_ab/$name/$unit.o" || objs="$objs $o"; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_ab/$name/_lipvq_hip.so $objs -ldl
    echo "built build_ab/$name/_lipvq_hip.so [$flags]" ) &
done
wait
