#!/bin/bash
# Dev tool (GPU box or here): build variant libraries side by side: scripts/ab_build.sh name "<extra -D flags>" [name2 "<flags2>" ...]
# -> build_ab/<name>/_lipvq_hip.so ; run with LIPVQ_HIP_LIBRARY=build_ab/<name>/_lipvq_hip.so (variants may compute wrong results by construction)
set -e
cd "$(dirname "$0")/.."
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  mkdir -p build_ab/$name/obj
  for f in lipvq-vae_amd/csrc/*.hip; do
    b=$(basename $f .hip)
    if [ "$b" = "lipvq_fused" ] || [ "$b" = "lipvq_screen" ] || [ ! -f build_ab/common/$b.o ]; then
      out=build_ab/$name/obj/$b.o
      [ "$b" = "lipvq_fused" ] || [ "$b" = "lipvq_screen" ] || { mkdir -p build_ab/common; out=build_ab/common/$b.o; }
      /opt/rocm/bin/hipcc $BASE $flags -c -o $out $f &
    fi
  done
  wait
  objs=""
  for f in lipvq-vae_amd/csrc/*.hip; do b=$(basename $f .hip); if [ -f build_ab/$name/obj/$b.o ]; then objs="$objs build_ab/$name/obj/$b.o"; else objs="$objs build_ab/common/$b.o"; fi; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_ab/$name/_lipvq_hip.so $objs -ldl
  echo "built build_ab/$name/_lipvq_hip.so [$flags]"
done
