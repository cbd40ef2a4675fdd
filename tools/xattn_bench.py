"""Cross-attention core micro-benchmark (BASELINE config 2: T=4096, E=1024, Lk=33), SURVEY.md 8d hygiene:
>= 100 back-to-back launches over a rotating set of buffers totalling > 512 MB ("cold": HBM) and over one
buffer pair ("warm": Infinity Cache / L2).  Algorithmic bytes per clip = 8*E (q in + ctx out) fp32.
Prints one JSON line."""
import ctypes, importlib, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(T, E, heads, Lk=33, B=1, reps=100):
    pkg = importlib.import_module('cvpr2025-decafnet_amd')
    lib = pkg._lib.lib()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rows = B * T
    per_set = 2 * rows * E * 4
    nsets = max(2, (600 * 2 ** 20) // per_set + 1)
    qs = [torch.randn(rows, E, device='cuda') for _ in range(nsets)]
    os_ = [torch.empty(rows, E, device='cuda') for _ in range(nsets)]
    k, v = torch.randn(B * Lk, E, device='cuda'), torch.randn(B * Lk, E, device='cuda')
    m = torch.ones(B * Lk, dtype=torch.bool, device='cuda')
    out = {}
    for mode in ('cold', 'warm'):
        for i in range(40):                      # long enough for the clocks to settle (the first ~20 launches run slow)
            lib.dcf_op_xattn(P(qs[i % nsets]), P(k), P(v), P(m), P(os_[i % nsets]), B, T, Lk, E, heads, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            j = i % nsets if mode == 'cold' else 0
            lib.dcf_op_xattn(P(qs[j]), P(k), P(v), P(m), P(os_[j]), B, T, Lk, E, heads, st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        byts = 8.0 * E * rows + 2 * 4.0 * B * Lk * E
        out[mode] = {'us': us, 'GBps': byts / us / 1e3, 'frac_hbm_peak': byts / us / 1e3 / 8000.0,
                     'tflops': 4.0 * E * Lk * rows / us / 1e6}
    return out


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'config2':        # only BASELINE config 2 (for the PMC passes)
        print(json.dumps(run(4096, 1024, 16, B=8)))
        sys.exit(0)
    res = {}
    for (T, E, h, B) in [(4096, 1024, 16, 1), (4096, 1024, 4, 1), (16384, 256, 4, 1), (16384, 256, 4, 8), (4096, 1024, 16, 8)]:
        res[f'T{T}_E{E}_h{h}_B{B}'] = run(T, E, h, B=B)
    print(json.dumps(res))
