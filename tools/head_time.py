"""A head as one kernel (csrc/head_chain.hip) against the GEMM / LayerNorm / output-convolution launches on the same rows:
python tools/head_time.py [C] [rows]; with a DCF_HC_STAMP build of the library also prints the in-kernel cycle shares."""
import ctypes
import importlib
import math
import os
import sys

import torch

sys.path.insert(0, os.environ.get('DCF_PKG_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module('cvpr2025-decafnet_amd')
lib = pkg._lib.lib()
P = pkg._lib.ptr


def main():
    C = int(sys.argv[1]) if len(sys.argv) > 1 else 288
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 261120
    NO, B = 2, 8
    T = rows // B
    g = torch.Generator().manual_seed(1)
    X = torch.randn(B * T, C, generator=g).cuda()
    mask = torch.ones(B, T, dtype=torch.uint8).cuda()
    W1 = (torch.randn(C, C, 3, generator=g) / math.sqrt(3 * C)).cuda()
    W2 = (torch.randn(C, C, 3, generator=g) / math.sqrt(3 * C)).cuda()
    lw, lb = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
    Wo = (torch.randn(NO, C, 3, generator=g) / math.sqrt(3 * C)).cuda()
    bo = torch.randn(NO, generator=g).cuda()
    out = torch.empty(B * T, NO, device='cuda')
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for chain in (0, 1):
        ts = []
        for it in range(5):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            pkg._lib.check(lib.dcf_op_head(P(X), P(mask), P(W1), P(lw), P(lb), P(W2), P(lw), P(lb), P(Wo), P(bo), P(out), B, T, C, NO, 1.0, chain, st))
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        print(f'C={C} rows={B * T} chain={chain}: whole op (with weight repacking) min {min(ts[1:]):.1f} us', flush=True)
    if hasattr(lib, 'dcf_debug_hc_stamps'):
        outv = (ctypes.c_ulonglong * 8)()
        lib.dcf_debug_hc_stamps.restype = ctypes.c_int
        assert lib.dcf_debug_hc_stamps(outv) == 0
        names = ['wait vmcnt(0)', 'barrier', 'MFMA stages', 'tap combine', 'LayerNorm + split', 'prologue (X load + split)', 'output conv + store', '-']
        tot = sum(outv)
        print(f'stamps: total {tot} cycles, {2 * 2 * (C // 32)} stages')
        for n, v in zip(names, outv):
            if v:
                print(f'   {n:28s} {v:9d}  {100.0 * v / tot:5.1f} %')


if __name__ == '__main__':
    main()
