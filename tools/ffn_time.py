"""FFN as one kernel (csrc/ffn_chain.hip) against the GEMM pair on the same rows: python tools/ffn_time.py [rows ...]
Times the kernels alone (weight split and scratch allocation happen inside dcf_op_ffn, so the per-call time is taken from the
library's per-launch profile, dcf_profile_enable)."""
import ctypes
import importlib
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('cvpr2025-decafnet_amd')
lib = pkg._lib.lib()
P = pkg._lib.ptr


def main():
    rows = [int(a) for a in sys.argv[1:]] or [131072, 65536, 32768, 16384]
    E = 256
    g = torch.Generator().manual_seed(1)
    W1 = (torch.randn(4 * E, E, generator=g) / math.sqrt(E)).cuda()
    b1 = (torch.randn(4 * E, generator=g) * 0.3).cuda()
    W2 = (torch.randn(E, 4 * E, generator=g) / math.sqrt(4 * E)).cuda()
    b2 = (torch.randn(E, generator=g) * 0.3).cuda()
    ls = torch.randn(E, generator=g).cuda()
    lw, lb = (torch.rand(E, generator=g) + 0.5).cuda(), (torch.randn(E, generator=g) * 0.5).cuda()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for M in rows:
        X = torch.randn(M, E, generator=g).cuda()
        mask = (torch.rand(M, generator=g) > 0.1).to(torch.uint8).cuda()
        C = torch.empty(M, E, device='cuda')
        for with_ln in (1, 0):
            for chain in (0, 2, 3):      # GEMM pair, four-wave kernel, eight-wave (producer / consumer) kernel
                ts = []
                for it in range(6):
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    pkg._lib.check(lib.dcf_op_ffn(P(X), P(lw) if with_ln else None, P(lb) if with_ln else None, P(W1), P(b1), P(W2), P(b2), P(ls),
                                                  P(mask), P(C), None, M, E, chain, st))
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                print(f'M={M} ln={with_ln} chain={chain}: whole op (with weight split / LayerNorm or row statistics) min {min(ts[1:]):.1f} us', flush=True)


if __name__ == '__main__':
    main()
