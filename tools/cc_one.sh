#!/bin/bash
# compile one kernel file of csrc/ alone with the resource-usage remarks and keep the ISA: tools/cc_one.sh ffn_chain
# -> /tmp/<name>.o, /tmp/<name>-hip-amdgcn-amd-amdhsa-gfx950.s, resource table on stdout
set -e
n=$1
cd /root/repo/cvpr2025-decafnet_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -fno-slp-vectorize \
  -Rpass-analysis=kernel-resource-usage -save-temps=obj -c $n.hip -o /tmp/$n.o 2> /tmp/res_$n.txt || { grep -E "error" -A4 /tmp/res_$n.txt | head -40; exit 1; }
grep -E "warning" -A3 /tmp/res_$n.txt | head -20 || true
python /root/repo/tools/kernel_resources.py /tmp/res_$n.txt
