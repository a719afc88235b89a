#!/bin/bash
# Dev tool (GPU box): rebuild the library with extra -D flags and time each build with scripts/measure_screen.py.
# Usage: bash scripts/ablate.sh "<flags A>" "<flags B>" ...   (ablation builds may compute WRONG results by construction)
set -e
cd "$(dirname "$0")/.."
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function"
for abl in "$@"; do
  make -s -C lipvq-vae_amd/csrc clean
  make -s -j8 -C lipvq-vae_amd/csrc FLAGS="$BASE $abl" 2>&1 | grep -E "error" || true
  echo "=== build: [$abl]"
  python scripts/${ABL_SCRIPT:-measure_screen.py} ${ABL_WL-cfg2} 2>&1 | grep -E "${ABL_GREP:-equal|screened|fused}" || true
done
make -s -C lipvq-vae_amd/csrc clean; make -s -j8 -C lipvq-vae_amd/csrc 2>&1 | grep error || true
