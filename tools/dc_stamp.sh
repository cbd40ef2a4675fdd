#!/bin/bash
# Diagnostic build of csrc/dec_chain.hip with in-kernel stamps (wave 0 of workgroup 1): private library copy under /tmp, one decoder
# layer (tools/dec_time.py), cycle shares of the kernel's segments.  GPU box only.  usage: tools/dc_stamp.sh [B] [T] [Lk]
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/dcstamp && mkdir -p /tmp/dcstamp && cp -r $R/cvpr2025-decafnet_amd /tmp/dcstamp/ && cd /tmp/dcstamp/cvpr2025-decafnet_amd
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -fno-slp-vectorize -DDCF_DC_STAMP $DC_EXTRA -c csrc/dec_chain.hip -o build/dec_chain.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o libdecafnet_hip.so build/*.o
DCF_PKG_ROOT=/tmp/dcstamp python3 $R/tools/dec_time.py "$@"
