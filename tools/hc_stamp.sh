#!/bin/bash
# Diagnostic build of csrc/head_chain.hip with in-kernel stamps (wave 0 of workgroup 1): private library copy under /tmp, the head
# operator alone (tools/head_time.py), cycle shares of the segments.  GPU box only.  usage: tools/hc_stamp.sh [C] [rows]
# HC_ABLATIONS="-DDCF_HC_NO_DMA -DDCF_HC_NO_LDS": further builds without the weight stream / without the fragment reads (timing only)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/hcstamp && mkdir -p /tmp/hcstamp && cp -r $R/cvpr2025-decafnet_amd /tmp/hcstamp/ && cd /tmp/hcstamp/cvpr2025-decafnet_amd
for abl in "" $HC_ABLATIONS; do
  echo "== build: -DDCF_HC_STAMP $abl"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -fno-slp-vectorize -DDCF_HC_STAMP $abl -c csrc/head_chain.hip -o build/head_chain.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o libdecafnet_hip.so build/*.o
  DCF_PKG_ROOT=/tmp/hcstamp python3 $R/tools/head_time.py "$@" | grep -v "chain=0"
done
