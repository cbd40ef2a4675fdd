#!/bin/bash
# Diagnostic build of csrc/ffn_chain.hip with in-kernel stamps (s_memtime brackets around the segments of an iteration, wave 0 of
# workgroup 0): builds a PRIVATE copy of the library under /tmp, runs one launch, prints the cycle shares.  GPU box only.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
P=$R/cvpr2025-decafnet_amd
mkdir -p /tmp/ffnstamp && cp -r $P /tmp/ffnstamp/ && cd /tmp/ffnstamp/cvpr2025-decafnet_amd
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -fno-slp-vectorize -DDCF_FFN_STAMP -c csrc/ffn_chain.hip -o build/ffn_chain.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o libdecafnet_hip.so build/*.o
cp -r $R/tools /tmp/ffnstamp/ 2>/dev/null || true
cd /tmp/ffnstamp && python3 - "$@" <<'PY'
import ctypes, importlib, math, sys, torch
sys.path.insert(0, '/tmp/ffnstamp')
pkg = importlib.import_module('cvpr2025-decafnet_amd')
lib = pkg._lib.lib(); P = pkg._lib.ptr
M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
E = 256
g = torch.Generator().manual_seed(1)
W1 = (torch.randn(4 * E, E, generator=g) / 16).cuda(); b1 = torch.randn(4 * E, generator=g).cuda()
W2 = (torch.randn(E, 4 * E, generator=g) / 32).cuda(); b2 = torch.randn(E, generator=g).cuda()
X = torch.randn(M, E, generator=g).cuda(); C = torch.empty(M, E, device='cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
names = ['wait vmcnt(0)', 'barrier', 'DMA issue', 'compute', 'epilogue', 'prologue (X load + split)', '-', '-']
for rep in range(3):
    pkg._lib.check(lib.dcf_op_ffn(P(X), None, None, P(W1), P(b1), P(W2), P(b2), None, None, P(C), None, M, E, 1, st))
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 8)()
    lib.dcf_debug_ffn_stamps.restype = ctypes.c_int
    assert lib.dcf_debug_ffn_stamps(out) == 0
    tot = sum(out)
    print(f'launch {rep}: total {tot} cycles (s_memtime), 34 iterations')
    for n, v in zip(names, out):
        if v: print(f'   {n:28s} {v:9d}  {100.0 * v / tot:5.1f} %   per iteration {v / 34:8.1f}')
PY
