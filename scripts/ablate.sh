#!/bin/bash
# Dev tool (GPU box): rebuild the library with parts of the screening kernels disabled and time each build.
# Results of ablated builds are WRONG by construction; only the timings are of interest.
set -e
cd "$(dirname "$0")/.."
export LQ_NO_USAGE=1
for abl in "-DLQ_ABL_CERT_ALL" "-DLQ_ABL_CERT_ALL -DLQ_ABL_NOLOOP" "-DLQ_ABL_CERT_ALL -DLQ_ABL_NOLOOP -DLQ_ABL_NOMERGE -DLQ_ABL_NOGATHER" "-DLQ_ABL_CERT_ALL -DLQ_ABL_NOTRACK"; do
  make -s -C lipvq-vae_amd/csrc clean
  make -s -j8 -C lipvq-vae_amd/csrc FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function $abl" 2>&1 | grep -E "error" || true
  echo "=== ablation: [$abl]"
  python scripts/measure_screen.py cfg2 2>&1 | grep -E "screened|fused" || true
done
make -s -C lipvq-vae_amd/csrc clean; make -s -j8 -C lipvq-vae_amd/csrc 2>&1 | grep error || true
