"""Reproducer for the sporadic miscompute of the row-statistics ("LayerNorm carried between two GEMMs") path that round 3
sidestepped with -fno-slp-vectorize (profiles/r03_notes.md section 1).

    python tools/micro/pkfma_repro.py [--reps 60] [--shape M,N1,K1,N2]

Runs dcf_op_linear_ln_carry REPS times on the same inputs with whatever library DCF_LIB_PATH names (default: the in-tree
build), compares every run bit for bit with the first one and with fp64, and prints where the differing elements sit in the
consumer kernel's epilogue: (row % 32, column) -> read-back lane = (row % 8) * 8 + (column % 32) / 4, component = column % 4,
quarter q = (row % 32) / 8.  A race or a hazard shows up as a changing set of elements; a deterministic miscompile as a fixed one.

Build the suspect variant first (on the build machine; the .so travels with the snapshot):
    python -c "import importlib; b = importlib.import_module('cvpr2025-decafnet_amd.build'); \
               print(b.build(variant='slp', drop={'gemm_bf16s.hip': ['-fno-slp-vectorize']}))"
"""
import argparse
import collections
import ctypes
import importlib
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=60)
    ap.add_argument('--shape', default='65600,256,256,1024')
    ap.add_argument('--res', type=int, default=1)
    ap.add_argument('--gelu', type=int, default=1)
    ap.add_argument('--load', type=int, default=0, help='run a bandwidth hog on a second stream beside the kernels')
    a = ap.parse_args()
    M, N1, K1, N2 = (int(v) for v in a.shape.split(','))
    pkg = importlib.import_module('cvpr2025-decafnet_amd')
    lib = pkg._lib.lib()
    print('library:', pkg._lib.SO_PATH)
    g = torch.Generator().manual_seed(M + N1 + N2)
    A = torch.randn(M, K1, generator=g)
    W1 = torch.randn(N1, K1, generator=g) / math.sqrt(K1)
    b1 = torch.randn(N1, generator=g) * 0.3
    R = torch.randn(M, N1, generator=g) if a.res else None
    lw, lb = torch.rand(N1, generator=g) + 0.5, torch.randn(N1, generator=g) * 0.5
    W2 = torch.randn(N2, N1, generator=g) / math.sqrt(N1)
    b2 = torch.randn(N2, generator=g) * 0.3
    x = A.double() @ W1.double().t() + b1.double()
    if a.res:
        x = x + R.double()
    mu = x.mean(1, keepdim=True)
    ln = (x - mu) / torch.sqrt(((x - mu) ** 2).mean(1, keepdim=True) + 1e-5) * lw.double() + lb.double()
    y = ln @ W2.double().t() + b2.double()
    if a.gelu:
        y = torch.nn.functional.gelu(y)
    d = {k: v.cuda() for k, v in dict(A=A, W1=W1, b1=b1, lw=lw, lb=lb, W2=W2, b2=b2).items()}
    Rd = R.cuda() if a.res else None
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    X = torch.empty(M, N1, device='cuda')
    Y = torch.empty(M, N2, device='cuda')
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    hog = None
    if a.load:
        hog_s = torch.cuda.Stream()
        hog = (torch.empty(64 << 20, device='cuda'), torch.empty(64 << 20, device='cuda'), hog_s)
    first_x = first_y = None
    bad_runs = 0
    where = collections.Counter()
    for rep in range(a.reps):
        X.fill_(float('nan'))
        Y.fill_(float('nan'))
        if hog is not None:
            with torch.cuda.stream(hog[2]):
                for _ in range(4):
                    hog[1].copy_(hog[0])
        pkg._lib.check(lib.dcf_op_linear_ln_carry(P(d['A']), P(d['W1']), P(d['b1']), P(Rd), P(d['lw']), P(d['lb']), P(d['W2']), P(d['b2']),
                                                  P(X), P(Y), M, N1, K1, N2, a.gelu, 16, st))
        torch.cuda.synchronize()
        if first_y is None:
            first_x, first_y = X.clone(), Y.clone()
            ex = float((X.cpu().double() - x).abs().max())
            ey = float((Y.cpu().double() - y).abs().max())
            print(f'run 0 against fp64: max |dX| = {ex:.3e}, max |dY| = {ey:.3e}')
            wrong = ((Y.cpu().double() - y).abs() > 2e-5 + 2e-5 * y.abs()).nonzero()
            if len(wrong):
                print(f'run 0: {len(wrong)} elements of Y outside 2e-5')
            continue
        dx = (X.view(torch.int32) != first_x.view(torch.int32)).nonzero()
        dy = (Y.view(torch.int32) != first_y.view(torch.int32)).nonzero()
        if len(dx) or len(dy):
            bad_runs += 1
            print(f'run {rep}: {len(dx)} elements of X and {len(dy)} of Y differ from run 0')
            for r, c in dy[:12].tolist():
                lane = (r % 8) * 8 + (c % 32) // 4
                print(f'   Y[{r}][{c}]  row%128={r % 128} q={(r % 32) // 8} lane={lane} comp={c % 4}  got {float(Y[r, c]):.7g} first {float(first_y[r, c]):.7g} fp64 {float(y[r, c]):.7g}')
            for r, c in dy.tolist():
                where[((r % 8) * 8 + (c % 32) // 4, c % 4)] += 1
    print(f'{bad_runs} of {a.reps - 1} repeats differ from the first run')
    if where:
        lanes = collections.Counter()
        comps = collections.Counter()
        for (lane, comp), n in where.items():
            lanes[lane] += n
            comps[comp] += n
        print('by read-back lane:', sorted(lanes.items()))
        print('by component:', sorted(comps.items()))
    return 1 if bad_runs else 0


if __name__ == '__main__':
    sys.exit(main())
