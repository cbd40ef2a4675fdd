#!/bin/bash
# A/B of two builds of the cross-attention core at BASELINE config 2 inside one GPU call, alternating (boxes differ by several per cent).
#   tools/xattn_ab.sh <variant name under cvpr2025-decafnet_amd/build/variants> [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}
V=${1:-xdirect}
N=${2:-3}
cd "$R" || exit 1
mkdir -p gpurun_out
test -f "cvpr2025-decafnet_amd/build/variants/$V/libdecafnet_hip.so" || { echo "no variant $V"; exit 1; }
for i in $(seq "$N"); do
  echo "product  $(python3 tools/xattn_bench.py config2 2>/dev/null)"
  echo "$V  $(DCF_LIB_PATH=$R/cvpr2025-decafnet_amd/build/variants/$V/libdecafnet_hip.so python3 tools/xattn_bench.py config2 2>/dev/null)"
done
