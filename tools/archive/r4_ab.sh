#!/bin/bash
# A/B of development libraries on the 8-chain attached run:  tools/r4_ab.sh <lib tag> [<lib tag> ...]   (ENVS="A=1 B=2" for extra settings)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for tag in "$@"; do
  echo "== $tag $ENVS"
  env $ENVS HML_LIBRARY=$ROOT/hammlet_amd/libhammlet_hip_k5$tag.so python tools/multi_chain.py ${CHAINS:-8} 1000 c3_1e8_k5_dynamic attached
done
