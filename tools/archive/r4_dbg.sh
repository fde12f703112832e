#!/bin/bash
# phase stamps of the many-chain block kernel (HML_FUSED_DEBUG=2) + timing, development library <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for tag in "$@"; do
echo "== $tag"
HML_LIBRARY=$ROOT/hammlet_amd/libhammlet_hip_k5$tag.so python tools/multi_chain.py 8 1000 c3_1e8_k5_dynamic attached
HML_FUSED_DEBUG=2 HML_LIBRARY=$ROOT/hammlet_amd/libhammlet_hip_k5$tag.so python tools/multi_chain.py 8 100 c3_1e8_k5_dynamic attached 2>&1 | grep -A5 "fused-many dbg" | tail -6
done
