#!/bin/bash
# Dev tool: build the library of a git revision (default HEAD) into build_ab/<name>/_lipvq_hip.so for same-box A/B runs
# against the working tree: scripts/ab_head.sh [name] [rev] ["extra -D flags"].  Run with LIPVQ_HIP_LIBRARY=build_ab/<name>/_lipvq_hip.so.
set -e
cd "$(dirname "$0")/.."
name=${1:-base}; rev=${2:-HEAD}; extra=${3:-}
tmp=$(mktemp -d)
git archive "$rev" lipvq-vae_amd/csrc include | tar -x -C "$tmp"
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function -fno-slp-vectorize"
make -s -j8 -C "$tmp/lipvq-vae_amd/csrc" FLAGS="$BASE $extra" >/dev/null
mkdir -p build_ab/$name
cp "$tmp/lipvq-vae_amd/_lipvq_hip.so" build_ab/$name/_lipvq_hip.so
rm -rf "$tmp"
echo "built build_ab/$name/_lipvq_hip.so from $rev [$extra]"
