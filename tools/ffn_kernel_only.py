"""helper of tools/ffn_pair_probe.sh: a few launches of the two FFN kernels of the library copy under argv[1] (run under rocprofv3)"""
import ctypes, importlib, math, os, sys, torch
sys.path.insert(0, sys.argv[1])
pkg = importlib.import_module('cvpr2025-decafnet_amd')
lib = pkg._lib.lib(); P = pkg._lib.ptr
M = int(sys.argv[2]); E = 256
g = torch.Generator().manual_seed(1)
W1 = (torch.randn(4 * E, E, generator=g) / 16).cuda(); b1 = torch.randn(4 * E, generator=g).cuda()
W2 = (torch.randn(E, 4 * E, generator=g) / 32).cuda(); b2 = torch.randn(E, generator=g).cuda()
X = torch.randn(M, E, generator=g).cuda(); C = torch.empty(M, E, device='cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for chain in (2, 3):
    for _ in range(7):
        pkg._lib.check(lib.dcf_op_ffn(P(X), None, None, P(W1), P(b1), P(W2), P(b2), None, None, P(C), None, M, E, chain, st))
        torch.cuda.synchronize()
