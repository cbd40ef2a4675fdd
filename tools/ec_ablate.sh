#!/bin/bash
# Ablation builds of csrc/enc_chain.hip (timing only, results are garbage): private library copy under /tmp, one encoder layer
# (tools/enc_time.py).  GPU box only.  usage: EC_EXTRA="-DENC_NO_VGATHER" tools/ec_ablate.sh [B] [T] [stride]
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/ecabl && mkdir -p /tmp/ecabl && cp -r $R/cvpr2025-decafnet_amd /tmp/ecabl/ && cd /tmp/ecabl/cvpr2025-decafnet_amd
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -fno-slp-vectorize $EC_EXTRA -c csrc/enc_chain.hip -o build/enc_chain.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o libdecafnet_hip.so build/*.o
DCF_PKG_ROOT=/tmp/ecabl python3 $R/tools/enc_time.py "$@"
