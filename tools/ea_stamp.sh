#!/bin/bash
# Diagnostic build of csrc/enc_chain.hip with in-kernel stamps in k_enc_attn (wave 0 of workgroup 1): private library copy under /tmp,
# one encoder layer (tools/enc_time.py), cycle shares of the kernel's segments.  GPU box only.  usage: tools/ea_stamp.sh [B] [T] [stride]
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/eastamp && mkdir -p /tmp/eastamp && cp -r $R/cvpr2025-decafnet_amd /tmp/eastamp/ && cd /tmp/eastamp/cvpr2025-decafnet_amd
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -fno-slp-vectorize -DDCF_EA_STAMP $EA_EXTRA -c csrc/enc_chain.hip -o build/enc_chain.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o libdecafnet_hip.so build/*.o
DCF_PKG_ROOT=/tmp/eastamp python3 $R/tools/enc_time.py "$@"
