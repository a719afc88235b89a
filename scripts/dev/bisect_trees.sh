#!/bin/bash
# Dev tool (container): full source trees of past revisions, each built in place, for same-box bisects of a timing regression:
#   scripts/dev/bisect_trees.sh rev1 rev2 ...  ->  build_ab/trees/<rev>/ (package + its own _lipvq_hip.so)
# then on the GPU box: for t in build_ab/trees/*; do python scripts/dev/bisect_measure.py $t; done
set -e
cd "$(dirname "$0")/../.."
for rev in "$@"; do
  d=build_ab/trees/$rev
  if [ -f $d/lipvq-vae_amd/_lipvq_hip.so ]; then echo "have $d"; continue; fi
  rm -rf $d; mkdir -p $d
  git archive "$rev" lipvq-vae_amd include lipvq_vae_amd.py | tar -x -C $d
  make -s -j${JOBS:-4} -C $d/lipvq-vae_amd/csrc > $d/build.log 2>&1 || { echo "build of $rev failed"; tail -5 $d/build.log; continue; }
  rm -rf $d/lipvq-vae_amd/csrc/build
  echo "built $d"
done
