"""HIP kernels vs the CPU oracle, operator by operator (through the C ABI).  GPU only."""
import ctypes
import math
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import Golden, ROOT, load_pkg

sys.path.insert(0, ROOT)
from oracle import decafnet_ref as R  # noqa: E402
from oracle import nms_oracle  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def L():
    pkg = load_pkg()
    lib = pkg._lib.lib()          # raises if the .so is missing: no silent fallback
    return pkg, lib


_KEEP = []


def P(t):
    """device address; the tensor is kept alive (temporaries like ``x.cuda()`` would otherwise be freed and
    their memory reused before the kernel has run)"""
    _KEEP.append(t)
    if len(_KEEP) > 256:
        torch.cuda.synchronize()
        del _KEEP[:128]
    return ctypes.c_void_p(t.data_ptr())


def st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def tok(x):
    """(bs, C, T) channel-major -> token-major rows (bs*T, C) on the GPU"""
    return x.permute(0, 2, 1).reshape(-1, x.size(1)).contiguous().cuda()


def untok(y, bs, T):
    return y.view(bs, T, -1).permute(0, 2, 1).cpu()


@pytest.mark.parametrize('M,N,K,act', [(64, 256, 256, 0), (1000, 256, 1024, 0), (333, 1024, 256, 1), (130, 512, 256, 0),
                                         (77, 288, 864, 2), (50, 160, 128, 0), (33, 128, 128, 0), (200, 96, 64, 0),
                                         (129, 64, 32, 0), (65, 32, 32, 0), (4096, 256, 256, 1)])
def test_linear(L, M, N, K, act):
    pkg, lib = L
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    ref = A.double() @ W.double().t() + b.double()
    if act == 1:
        ref = F.gelu(ref)
    elif act == 2:
        ref = F.relu(ref)
    C = torch.empty(M, N, device='cuda')
    pkg._lib.check(lib.dcf_op_linear(P(A.cuda()), P(W.cuda()), P(b.cuda()), P(C), M, N, K, act, st()))
    torch.testing.assert_close(C.cpu().double(), ref, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize('nterms,tol', [(6, 2e-5), (16, 2e-5)])
@pytest.mark.parametrize('M,N,K,act', [(64, 256, 256, 0), (1000, 256, 1024, 1), (333, 1024, 256, 0), (77, 288, 864, 2),
                                         (50, 160, 128, 0), (200, 96, 64, 0), (129, 64, 32, 0), (65, 32, 32, 0), (4096, 512, 256, 0),
                                         (65600, 256, 256, 1), (32704, 288, 96, 0), (20000, 256, 128, 2),   # 128x256 / 128x96 / 64x256 tiles
                                         (33000, 1024, 256, 1), (40001, 288, 864, 2), (70000, 256, 1024, 0), (32768, 512, 64, 0)])
def test_linear_bf16_split(L, M, N, K, act, nterms, tol):
    """fp32-accurate GEMM on the 16-bit matrix cores (operand splitting, gemm_bf16s.hip: 6 = bf16x6, 16 = f16x3) vs fp64"""
    pkg, lib = L
    g = torch.Generator().manual_seed(M * 5 + N + nterms)
    A = torch.randn(M, K, generator=g) * 3
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    ref = A.double() @ W.double().t() + b.double()
    if act == 1:
        ref = F.gelu(ref)
    elif act == 2:
        ref = F.relu(ref)
    C = torch.empty(M, N, device='cuda')
    pkg._lib.check(lib.dcf_op_linear_split(P(A.cuda()), P(W.cuda()), P(b.cuda()), P(C), M, N, K, act, nterms, st()))
    torch.testing.assert_close(C.cpu().double(), ref, rtol=tol, atol=tol)
    if True:            # not worse than the native fp32 MFMA GEMM
        C32 = torch.empty(M, N, device='cuda')
        pkg._lib.check(lib.dcf_op_linear(P(A.cuda()), P(W.cuda()), P(b.cuda()), P(C32), M, N, K, act, st()))
        e6 = (C.cpu().double() - ref).abs().max()
        e32 = (C32.cpu().double() - ref).abs().max()
        assert e6 <= 2.0 * e32 + 1e-7, (float(e6), float(e32))


@pytest.mark.parametrize('M,K,relu,raw,N,nterms', [(28700, 96, 1, False, 256, 6), (32768, 256, 0, True, 256, 6), (28800, 768, 1, False, 256, 16)])
def test_linear_with_fused_layernorm(L, M, K, relu, raw, N, nterms):
    """GEMM with the channel LayerNorm (+ReLU) of the output row in its epilogue vs fp64 (last row tile partial)"""
    pkg, lib = L
    g = torch.Generator().manual_seed(M + K)
    A = torch.randn(M, K, generator=g) * 2
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b, lw, lb = torch.randn(N, generator=g), torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
    v = A.double() @ W.double().t() + b.double()
    mu = v.mean(1, keepdim=True)
    ref = (v - mu) / torch.sqrt(((v - mu) ** 2).mean(1, keepdim=True) + 1e-5) * lw.double() + lb.double()
    if relu:
        ref = F.relu(ref)
    C = torch.empty(M, N, device='cuda') if raw else None
    Y = torch.empty(M, N, device='cuda')
    pkg._lib.check(lib.dcf_op_linear_ln(P(A.cuda()), P(W.cuda()), P(b.cuda()), P(lw.cuda()), P(lb.cuda()),
                                        P(C) if raw else None, P(Y), M, N, K, relu, nterms, st()))
    torch.testing.assert_close(Y.cpu().double(), ref, rtol=5e-5, atol=5e-5)
    if raw:
        torch.testing.assert_close(C.cpu().double(), v, rtol=2e-5, atol=2e-5)


# (M, N1, K1, N2, residual, gelu, nterms, row mean): tile shapes 128x256 (M >= 65536), 64x256, 64x128, 64x64 on the producer side
# (1 / 1 / 2 / 4 partial-sum slots per row), the last row tile partial; a row mean of 3 sigma checks the one-pass variance
@pytest.mark.parametrize('M,N1,K1,N2,res,gelu,nterms,shift', [
    (65600, 256, 256, 1024, 1, 1, 16, 0.0), (28700, 256, 256, 1024, 1, 1, 16, 0.0), (14400, 256, 256, 1024, 0, 1, 16, 3.0),
    (8200, 256, 128, 256, 1, 0, 16, 0.0), (28700, 512, 256, 512, 0, 1, 6, 0.0), (16500, 128, 64, 512, 0, 0, 16, -2.0)])
def test_linear_layernorm_carried_as_row_statistics(L, M, N1, K1, N2, res, gelu, nterms, shift):
    """X = A W1^T + b1 (+ R) with (sum, sum of squares) per row on the side, then act(LayerNorm(X) W2^T + b2) from the raw X
    with the gain folded into W2 and (mean, rstd) applied in the epilogue -- against fp64 (dcf_op_linear_ln_carry)"""
    pkg, lib = L
    g = torch.Generator().manual_seed(M + N1 + N2)
    A = torch.randn(M, K1, generator=g)
    W1 = torch.randn(N1, K1, generator=g) / math.sqrt(K1)
    b1 = torch.randn(N1, generator=g) * 0.3 + shift
    R = torch.randn(M, N1, generator=g) if res else None
    lw, lb = torch.rand(N1, generator=g) + 0.5, torch.randn(N1, generator=g) * 0.5
    W2 = torch.randn(N2, N1, generator=g) / math.sqrt(N1)
    b2 = torch.randn(N2, generator=g) * 0.3
    x = A.double() @ W1.double().t() + b1.double()
    if res:
        x = x + R.double()
    mu = x.mean(1, keepdim=True)
    ln = (x - mu) / torch.sqrt(((x - mu) ** 2).mean(1, keepdim=True) + 1e-5) * lw.double() + lb.double()
    y = ln @ W2.double().t() + b2.double()
    if gelu:
        y = F.gelu(y)
    X = torch.empty(M, N1, device='cuda')
    Y = torch.empty(M, N2, device='cuda')
    d = {k: v.cuda() for k, v in dict(A=A, W1=W1, b1=b1, lw=lw, lb=lb, W2=W2, b2=b2).items()}     # alive until the synchronising .cpu() below
    Rd = R.cuda() if res else None
    pkg._lib.check(lib.dcf_op_linear_ln_carry(P(d['A']), P(d['W1']), P(d['b1']), P(Rd) if res else None, P(d['lw']), P(d['lb']),
                                              P(d['W2']), P(d['b2']), P(X), P(Y), M, N1, K1, N2, gelu, nterms, st()))
    torch.testing.assert_close(X.cpu().double(), x, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(Y.cpu().double(), y, rtol=2e-5, atol=2e-5)
    with pytest.raises(RuntimeError, match='k-sliced'):                  # small grids do not carry statistics: refused, not wrong
        pkg._lib.check(lib.dcf_op_linear_ln_carry(P(d['A']), P(d['W1']), P(d['b1']), None, P(d['lw']), P(d['lb']),
                                                  P(d['W2']), P(d['b2']), P(X), P(Y), 1024, N1, K1, N2, gelu, nterms, st()))


@pytest.mark.parametrize('M,N1,K1,N2,res,gelu', [(65600, 256, 256, 1024, 1, 1), (14400, 256, 256, 1024, 0, 1)])
def test_layernorm_carry_bit_identical_across_repeats(L, M, N1, K1, N2, res, gelu):
    """Determinism gate of the row-statistics path (dcf_op_linear_ln_carry): 30 repeats on the same inputs, X and Y bit for bit
    equal to the first run.  With the SLP-vectorised fold (v_pk_fma_f32 ... op_sel:[0,1,0]) every repeat differed in a few
    thousand elements (tools/micro/pkfma_repro.py, profiles/r04_pkfma_hazard.md)."""
    pkg, lib = L
    g = torch.Generator().manual_seed(M + N1 + N2)
    d = {k: v.cuda() for k, v in dict(A=torch.randn(M, K1, generator=g), W1=torch.randn(N1, K1, generator=g) / math.sqrt(K1),
                                      b1=torch.randn(N1, generator=g) * 0.3, R=torch.randn(M, N1, generator=g),
                                      lw=torch.rand(N1, generator=g) + 0.5, lb=torch.randn(N1, generator=g) * 0.5,
                                      W2=torch.randn(N2, N1, generator=g) / math.sqrt(N1), b2=torch.randn(N2, generator=g) * 0.3).items()}
    X = torch.empty(M, N1, device='cuda')
    Y = torch.empty(M, N2, device='cuda')
    first = None
    for rep in range(31):
        X.fill_(float('nan'))
        Y.fill_(float('nan'))
        pkg._lib.check(lib.dcf_op_linear_ln_carry(P(d['A']), P(d['W1']), P(d['b1']), P(d['R']) if res else None, P(d['lw']), P(d['lb']),
                                                  P(d['W2']), P(d['b2']), P(X), P(Y), M, N1, K1, N2, gelu, 16, st()))
        torch.cuda.synchronize()
        if first is None:
            first = (X.clone(), Y.clone())
            continue
        assert torch.equal(X.view(torch.int32), first[0].view(torch.int32)), f'repeat {rep}: X differs from the first run'
        assert torch.equal(Y.view(torch.int32), first[1].view(torch.int32)), f'repeat {rep}: Y differs from the first run'


# (B, T, C, NO, scale): both widths, both output counts, sequences shorter and longer than a 104-row tile, ragged masks with
# holes, several sequences back to back (their ends must not see each other)
@pytest.mark.parametrize('B,T,C,NO,scale', [(3, 100, 256, 1, 0.0), (2, 333, 288, 2, 1.7), (5, 64, 288, 1, 0.0), (1, 4000, 256, 2, 0.6)])
def test_head_chain_vs_fp64(L, B, T, C, NO, scale):
    """a whole head in one kernel (csrc/head_chain.hip: two k3 trunk layers with LayerNorm + ReLU and the output convolution,
    activations in registers) against fp64 torch and against the launches it replaces (head.py:53-64, :95-103)"""
    pkg, lib = L
    g = torch.Generator().manual_seed(B * 1000 + T + C + NO)
    X = torch.randn(B * T, C, generator=g)
    mask = torch.ones(B, T, dtype=torch.bool)
    for b in range(B):
        n = int(torch.randint(T // 2, T + 1, (1,), generator=g))
        mask[b, n:] = False
        if T > 40:
            mask[b, 17] = False                                  # a hole inside the valid part
    W1 = torch.randn(C, C, 3, generator=g) / math.sqrt(3 * C)
    W2 = torch.randn(C, C, 3, generator=g) / math.sqrt(3 * C)
    l1w, l1b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    l2w, l2b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    Wo = torch.randn(NO, C, 3, generator=g) / math.sqrt(3 * C)
    bo = torch.randn(NO, generator=g)
    mf = mask[:, None, :].double()
    x = X.double().view(B, T, C).transpose(1, 2)                 # (B, C, T)

    def ln(v, w, b):
        mu = v.mean(1, keepdim=True)
        return (v - mu) / torch.sqrt(((v - mu) ** 2).mean(1, keepdim=True) + 1e-5) * w.double()[None, :, None] + b.double()[None, :, None]

    y = torch.relu(ln(F.conv1d(x * mf, W1.double(), padding=1), l1w, l1b))
    y = torch.relu(ln(F.conv1d(y * mf, W2.double(), padding=1), l2w, l2b))
    o = F.conv1d(y * mf, Wo.double(), bo.double(), padding=1)
    if scale != 0.0:
        o = torch.relu(o * scale)
    ref = o.transpose(1, 2).reshape(B * T, NO)
    valid = mask.view(-1)
    d = {k: v.cuda() for k, v in dict(X=X, mask=mask.to(torch.uint8), W1=W1, W2=W2, l1w=l1w, l1b=l1b, l2w=l2w, l2b=l2b, Wo=Wo, bo=bo).items()}
    outs = {}
    for chain in (1, 0):
        out = torch.full((B * T + 1, NO), float('nan'), device='cuda')
        pkg._lib.check(lib.dcf_op_head(P(d['X']), P(d['mask']), P(d['W1']), P(d['l1w']), P(d['l1b']), P(d['W2']), P(d['l2w']), P(d['l2b']),
                                       P(d['Wo']), P(d['bo']), P(out), B, T, C, NO, scale, chain, st()))
        oc = out.cpu()
        assert torch.isnan(oc[B * T]).all(), 'wrote beyond the last row'
        assert torch.isfinite(oc[:B * T]).all()
        torch.testing.assert_close(oc[:B * T][valid].double(), ref[valid], rtol=3e-5, atol=3e-5)
        outs[chain] = oc[:B * T]
    torch.testing.assert_close(outs[1][valid], outs[0][valid], rtol=2e-5, atol=2e-5)
    with pytest.raises(RuntimeError, match='256 / 288'):
        pkg._lib.check(lib.dcf_op_head(P(d['X']), P(d['mask']), P(d['W1']), P(d['l1w']), P(d['l1b']), P(d['W2']), P(d['l2w']), P(d['l2b']),
                                       P(d['Wo']), P(d['bo']), P(out), 1, 8, 128, NO, scale, 1, st()))


# (M, LayerNorm in front, LayerScale, row mask, row statistics out, row mean): a partial last tile, fewer rows than one tile,
# one tile per CU and several rounds of tiles
@pytest.mark.parametrize('M,with_ln,with_ls,with_mask,with_stats,shift', [
    (128, 0, 0, 0, 0, 0.0), (200, 1, 1, 1, 0, 0.0), (33000, 1, 1, 1, 1, 0.0), (70000, 0, 1, 0, 1, 2.0), (40, 1, 0, 1, 0, -1.5),
    (131072 + 300, 1, 1, 1, 1, 0.5)])
def test_ffn_chain_vs_fp64(L, M, with_ln, with_ls, with_mask, with_stats, shift):
    """FFN in one kernel (csrc/ffn_chain.hip: both products transposed, the hidden activations in registers) against fp64
    and against the GEMM pair it replaces (blocks.py:535-538, 589-590).  chain 1 = the default kernel, 2 = the four-wave kernel,
    3 = the eight-wave kernel (producer / consumer wave pairs, persistent over its row tiles: the last case gives a workgroup
    more than four tiles and a ragged last one): the two kernels must agree bit for bit."""
    pkg, lib = L
    E = 256
    g = torch.Generator().manual_seed(M + with_ln)
    X = torch.randn(M, E, generator=g) + shift
    lw, lb = torch.rand(E, generator=g) + 0.5, torch.randn(E, generator=g) * 0.5
    W1 = torch.randn(4 * E, E, generator=g) / math.sqrt(E)
    b1 = torch.randn(4 * E, generator=g) * 0.3
    W2 = torch.randn(E, 4 * E, generator=g) / math.sqrt(4 * E)
    b2 = torch.randn(E, generator=g) * 0.3
    ls = torch.randn(E, generator=g)
    mask = torch.rand(M, generator=g) > 0.2
    x = X.double()
    xin = x
    if with_ln:
        mu = x.mean(1, keepdim=True)
        xin = (x - mu) / torch.sqrt(((x - mu) ** 2).mean(1, keepdim=True) + 1e-5) * lw.double() + lb.double()
    y = F.gelu(xin @ W1.double().t() + b1.double()) @ W2.double().t() + b2.double()
    if with_mask:
        y = y * mask[:, None].double()
    if with_ls:
        y = y * ls.double()
    ref = x + y
    d = {k: v.cuda() for k, v in dict(X=X, lw=lw, lb=lb, W1=W1, b1=b1, W2=W2, b2=b2, ls=ls, mask=mask.to(torch.uint8)).items()}
    out = {}
    for chain in (1, 2, 3, 0):
        if chain == 0 and with_stats and M < 8192:
            continue
        C = torch.full((M + 1, E), float('nan'), device='cuda')          # one guard row behind the last one
        S = torch.full((M, E // 64, 2), float('nan'), device='cuda') if with_stats else None
        pkg._lib.check(lib.dcf_op_ffn(P(d['X']), P(d['lw']) if with_ln else None, P(d['lb']) if with_ln else None, P(d['W1']), P(d['b1']),
                                      P(d['W2']), P(d['b2']), P(d['ls']) if with_ls else None, P(d['mask']) if with_mask else None,
                                      P(C), P(S) if with_stats else None, M, E, chain, st()))
        Cc = C.cpu()
        assert torch.isnan(Cc[M]).all(), 'wrote beyond the last row'
        torch.testing.assert_close(Cc[:M].double(), ref, rtol=2e-5, atol=2e-5)
        out[chain] = Cc[:M]
        if with_stats:
            Sc = S.cpu().double().sum(1)
            torch.testing.assert_close(Sc[:, 0], ref.sum(1), rtol=1e-4, atol=2e-3)
            torch.testing.assert_close(Sc[:, 1], (ref * ref).sum(1), rtol=1e-4, atol=2e-3)
        out[('S', chain)] = S.cpu() if with_stats else None
    assert torch.equal(out[2], out[3]) and torch.equal(out[1], out[3]), 'the four-wave and the eight-wave kernel differ'
    if with_stats:
        assert torch.equal(out[('S', 2)], out[('S', 3)])
    if 0 in out:
        tol = 1e-5 if with_ln else 2e-6      # without the LayerNorm fold: same products, same order, a few ulps at most
        torch.testing.assert_close(out[1], out[0], rtol=tol, atol=tol)
    with pytest.raises(RuntimeError, match='E = 256'):
        pkg._lib.check(lib.dcf_op_ffn(P(d['X']), None, None, P(d['W1']), P(d['b1']), P(d['W2']), P(d['b2']), None, None, P(C), None, 8, 128, 1, st()))


@pytest.mark.parametrize('ratio', [30.0, 300.0])
def test_ffn_chain_guards_the_one_pass_layernorm(L, ratio):
    """rows whose mean dwarfs their spread (|mean| / sigma = 30, 300): the LayerNorm in front of the FFN kernel rides as one-pass row
    statistics (var = E[x^2] - mean^2: relative error ~1e-7 mean^2 / var), the kernel flags such rows (common.h LN_ILL_RATIO, numerics
    bit 16) and the call repeats with the two-pass LayerNorm launch -- the FFN's contribution C - X must hold the fp64 reference at the
    usual tolerance (unguarded: 5e-5 .. 5e-3 off).  blocks.py:125-131 is two-pass."""
    pkg, lib = L
    E, M = 256, 20000
    g = torch.Generator().manual_seed(int(ratio))
    X = torch.randn(M, E, generator=g) + ratio * (torch.rand(M, 1, generator=g) * 0.2 + 0.9) * torch.where(torch.rand(M, 1, generator=g) > 0.5, 1.0, -1.0)
    lw, lb = torch.rand(E, generator=g) + 0.5, torch.randn(E, generator=g) * 0.5
    W1 = torch.randn(4 * E, E, generator=g) / math.sqrt(E)
    b1 = torch.randn(4 * E, generator=g) * 0.3
    W2 = torch.randn(E, 4 * E, generator=g) / math.sqrt(4 * E)
    b2 = torch.randn(E, generator=g) * 0.3
    x = X.double()
    mu = x.mean(1, keepdim=True)
    xin = (x - mu) / torch.sqrt(((x - mu) ** 2).mean(1, keepdim=True) + 1e-5) * lw.double() + lb.double()
    y = F.gelu(xin @ W1.double().t() + b1.double()) @ W2.double().t() + b2.double()
    d = {k: v.cuda() for k, v in dict(X=X, lw=lw, lb=lb, W1=W1, b1=b1, W2=W2, b2=b2).items()}
    for chain in (1, 2, 3):
        C = torch.empty(M, E, device='cuda')
        pkg._lib.check(lib.dcf_op_ffn(P(d['X']), P(d['lw']), P(d['lb']), P(d['W1']), P(d['b1']), P(d['W2']), P(d['b2']), None, None,
                                      P(C), None, M, E, chain, st()))
        # what fp32 itself allows at |x| ~ ratio, sigma = 1: the row mean (a sum of 256 values ~ratio) and x - mean are good to a few
        # ulp(ratio) whatever the algorithm -- the reference's fp32 LayerNorm as well -- and C = X + y is rounded at |X| ~ ratio
        got_y = C.cpu().double() - x
        ulp = 2.0 ** (math.floor(math.log2(ratio * 1.2)) - 23)
        torch.testing.assert_close(got_y, y, rtol=2e-5, atol=2e-5 + 8 * ulp)


def test_layernorm_carry_trips_the_guard_and_the_model_falls_back(L):
    """the same guard on a model: an encoder layer whose input rows have |mean| >> sigma raises numerics bit 16; set_ln_carry(False)
    sends every LayerNorm through its own two-pass launch and the flag stays clear"""
    import ctypes
    pkg, lib = L
    kw = dict(D=64, E=256, TE=64, text_in=32, n_levels=2, win=5, n_heads=4, sn=8, sratio=0.3, msf=True, norm=True,
              max_seq_len=128, text_layers=1, text_max_len=24)
    opt = pkg.config.make_opt(**kw)
    model = pkg.modeling.create_model(opt)
    sd = pkg.synth.make_state_dict({k: list(v.shape) for k, v in model.state_dict().items()}, 3)
    # a DC-heavy residual stream: the last embedding LayerNorm's bias lifts every channel of every clip by 40 (ReLU keeps it), so the
    # rows that reach ln_ffn of the first encoder layer -- carried as row statistics -- have |mean| ~ 40 at a spread below 1
    sd['vid_net.embd_norms.1.bias'] = sd['vid_net.embd_norms.1.bias'] + 40.0
    model.load_state_dict(sd)
    model = model.cuda().eval().requires_grad_(False)
    T = 20480
    inp = pkg.synth.make_inputs(64, T, T, 1, 32, 6, 4)
    tm = [model.encode_text(t[None].cuda(), torch.ones(1, 1, t.size(-1), dtype=torch.bool, device='cuda')) for t in inp['tokens']]
    args = (inp['vid'].cuda(), inp['shallow_vid'].cuda(), inp['vid_masks'].cuda(), tuple(t for t, _ in tm), inp['text_cls'].cuda(), tuple(m for _, m in tm))
    first = model(*args, eval=True)
    torch.cuda.synchronize()
    assert model.ln_carry
    # a plain model(...) caller never mistakes that forward for valid scores: its logits are NaN on the device (k_masks_out, bit 1 like bit 0)
    assert all(bool(torch.isnan(first[0][0][l]).all()) for l in range(2))
    with pytest.raises(RuntimeError, match='mean dwarfs its spread'):        # the next call into the model sees the flag of the first one
        model(*args, eval=True)
    assert not model.ln_carry, 'the model must have switched to the two-pass LayerNorm launches by itself'
    out = model(*args, eval=True)
    assert model.numerics_status(reset=True) & 16 == 0
    from oracle import decafnet_ref as R
    want = R.forward_eval(sd, opt.model, inp['vid'], inp['shallow_vid'], inp['vid_masks'], [t.cpu() for t, _ in tm], inp['text_cls'], [m.cpu() for _, m in tm])
    for l in range(2):
        torch.testing.assert_close(out[0][0][l].cpu(), want[0][0][l], rtol=2e-4, atol=2e-4)
    # GroundingEvaluator does not abort a run on that flag: it switches the model and REPEATS the affected videos (predict, and the three
    # modes of run); the proposals are those of a model that ran two-pass LayerNorms from the start
    ev = pkg.evaluator
    opt.model['max_vid_len'] = T
    g = torch.Generator().manual_seed(11)
    videos = [dict(vid=inp['vid'][0], shallow_vid=inp['shallow_vid'][0], text=tuple(inp['tokens']), text_cls=inp['text_cls'],
                   segment=torch.rand(1, 2, generator=g).sort(1).values * T * 16 / 30.0, fps=30.0, clip_stride=16, clip_size=32,
                   duration=T * 16 / 30.0 + 2) for _ in range(3)]
    want_res = ev.GroundingEvaluator(opt, model).predict(videos[0])          # (the model runs two-pass launches by now)
    assert len(want_res) == 1 and bool(torch.isfinite(want_res[0]['scores']).all()) and want_res[0]['scores'].numel() > 0

    class Log(ev.RecallCounter):
        def __init__(self):
            super().__init__((1, 5), (0.3, 0.5))
            self.log = []

        def update(self, results, targets):
            self.log.append([(r['segments'].cpu().clone(), r['scores'].cpu().clone()) for r in results])
            super().update(results, targets)

    for mode in (dict(), dict(batch_videos=2), dict(n_streams=2)):
        model.set_ln_carry(True)
        e = ev.GroundingEvaluator(opt, model)
        if not mode:
            res = e.predict(videos[0])
            assert not model.ln_carry and torch.equal(res[0]['scores'], want_res[0]['scores']) and torch.equal(res[0]['segments'], want_res[0]['segments'])
            model.set_ln_carry(True)
        c = e.run(videos, Log(), **mode)
        assert not model.ln_carry and len(c.log) == 3, mode
        for entry in c.log:
            if 'batch_videos' in mode:             # (two videos per forward take other kernels at this width: equal to round-off, not bit for bit)
                torch.testing.assert_close(entry[0][1], want_res[0]['scores'], rtol=1e-4, atol=1e-4)
                torch.testing.assert_close(entry[0][0], want_res[0]['segments'], rtol=1e-4, atol=1e-3)
            else:
                assert torch.equal(entry[0][1], want_res[0]['scores']) and torch.equal(entry[0][0], want_res[0]['segments']), mode
        assert model.numerics_status(reset=True) == 0


@pytest.mark.parametrize('nterms,tol', [(6, 2e-5), (16, 2e-5)])
@pytest.mark.parametrize('M,N,K', [(256, 256, 512), (1000, 128, 64), (4096, 256, 1024), (16384, 256, 96), (20, 128, 32)])
def test_linear_channel_major_split(L, M, N, K, nterms, tol):
    """channel-major A on the bf16-split path (vid_map): transposed LDS staging of packed (k, k+1) pairs"""
    pkg, lib = L
    g = torch.Generator().manual_seed(M + N + K + nterms)
    X = torch.randn(K, M, generator=g) * 2          # the reference's (C, T) layout
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    ref = X.double().t() @ W.double().t() + b.double()
    C = torch.empty(M, N, device='cuda')
    pkg._lib.check(lib.dcf_op_linear_cm_split(P(X.cuda()), P(W.cuda()), P(b.cuda()), P(C), M, N, K, nterms, st()))
    torch.testing.assert_close(C.cpu().double(), ref, rtol=tol, atol=tol)


@pytest.mark.parametrize('M,N,K', [(256, 256, 512), (1000, 128, 64), (4096, 256, 1024), (250, 64, 32)])
def test_linear_channel_major(L, M, N, K):
    pkg, lib = L
    g = torch.Generator().manual_seed(M + N + K)
    X = torch.randn(K, M, generator=g)             # the reference's (C, T) layout
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    ref = X.double().t() @ W.double().t()
    C = torch.empty(M, N, device='cuda')
    pkg._lib.check(lib.dcf_op_linear_cm(P(X.cuda()), P(W.cuda()), None, P(C), M, N, K, st()))
    torch.testing.assert_close(C.cpu().double(), ref, rtol=1e-5, atol=2e-5)


def test_conv3_golden(L):
    pkg, lib = L
    ops = Golden('ops.npz')
    x, mask = ops.t('x'), ops.t('mask')
    w = ops.t('conv_k3/w/conv.weight')
    bs, Cc, T = x.shape
    Y = torch.empty(bs * T, Cc, device='cuda')
    pkg._lib.check(lib.dcf_op_conv3(P(tok(x)), P(mask.reshape(-1).contiguous().cuda()), P(w.contiguous().cuda()), P(Y), bs, T, Cc, Cc, st()))
    torch.testing.assert_close(untok(Y, bs, T), ops.t('conv_k3/y'), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('nterms', [16, 6])
@pytest.mark.parametrize('B,T,Cin,N', [(2, 300, 64, 64), (3, 11008, 256, 256), (2, 16400, 288, 288), (1, 33000, 32, 256)])
def test_conv3_split_vs_fp64(L, B, T, Cin, N, nterms):
    """MaskedConv1D k3 (blocks.py:63-106) on the split-operand GEMM path (three taps through the neighbour flags: sequence
    boundaries and padded tails inside the row tiles, 64- and 128-row tile kernels) against an fp64 convolution"""
    pkg, lib = L
    g = torch.Generator().manual_seed(B * T + Cin)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(N, Cin, 3, generator=g) / math.sqrt(3 * Cin)
    mask = torch.ones(B, 1, T, dtype=torch.bool)
    mask[0, :, int(T * 0.9):] = False
    mask[-1, :, T - 3:] = False
    ref = F.conv1d((x * mask).double(), w.double(), padding=1)
    Y = torch.empty(B * T, N, device='cuda')
    pkg._lib.check(lib.dcf_op_conv3_split(P(tok(x)), P(mask.reshape(-1).contiguous().cuda()), P(w.contiguous().cuda()), P(Y), B, T, Cin, N, nterms, st()))
    torch.testing.assert_close(untok(Y, B, T).double(), ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('C', [32, 128, 256, 288, 1024])
def test_layernorm(L, C):
    pkg, lib = L
    g = torch.Generator().manual_seed(C)
    x = torch.randn(1, C, 77, generator=g) * 3 + 1
    w, b = torch.randn(C, 1, generator=g), torch.randn(C, 1, generator=g)
    ref = R.channel_layer_norm(x, w, b)
    Y = torch.empty(77, C, device='cuda')
    pkg._lib.check(lib.dcf_op_layernorm(P(tok(x)), P(w.cuda()), P(b.cuda()), P(Y), 77, C, 0, st()))
    torch.testing.assert_close(untok(Y, 1, 77), ref, rtol=1e-5, atol=1e-5)
    pkg._lib.check(lib.dcf_op_layernorm(P(tok(x)), None, None, P(Y), 77, C, 1, st()))
    torch.testing.assert_close(untok(Y, 1, 77), F.relu(R.channel_layer_norm(x)), rtol=1e-5, atol=1e-5)


def xattn_ref(q, k, v, kvmask, heads):
    """q (B,T,C), k/v (B,Lk,C) token-major; MaskedMHA global core, libs/modeling/blocks.py:375-389"""
    B, T, C = q.shape
    d = C // heads
    s = 1.0 / math.sqrt(math.sqrt(d))
    qh = q.view(B, T, heads, d).transpose(1, 2) * s
    kh = k.view(B, -1, heads, d).transpose(1, 2) * s
    vh = v.view(B, -1, heads, d).transpose(1, 2)
    att = (qh @ kh.transpose(2, 3)).masked_fill(~kvmask[:, None, None, :], float('-inf'))
    return (F.softmax(att, -1) @ vh).transpose(1, 2).reshape(B, T, C)


@pytest.mark.parametrize('B,T,Lk,C,heads', [(1, 300, 33, 256, 4), (2, 70, 17, 128, 4), (1, 64, 9, 64, 2), (1, 257, 33, 1024, 4),
                                             (1, 100, 33, 1024, 16), (3, 40, 6, 32, 4),
                                             # head dims 128 and 64 with ragged ends: the context rows leave through the staged LDS tile
                                             # (whole rows per store; the last row group is partly beyond T)
                                             (1, 130, 33, 512, 4), (2, 77, 20, 256, 2), (2, 1000, 48, 256, 4), (1, 17, 5, 128, 2),
                                             # 64 keys at head dims 128 / 256: 100 / 136 KB of LDS (the launch raises the kernel's limit)
                                             (1, 90, 64, 512, 4), (1, 70, 64, 1024, 4)])
def test_xattn_core(L, B, T, Lk, C, heads):
    pkg, lib = L
    g = torch.Generator().manual_seed(T + C)
    q, k, v = torch.randn(B, T, C, generator=g), torch.randn(B, Lk, C, generator=g), torch.randn(B, Lk, C, generator=g)
    m = torch.ones(B, Lk, dtype=torch.bool)
    m[-1, Lk // 2:] = False
    ref = xattn_ref(q, k, v, m, heads)
    O = torch.empty(B * T, C, device='cuda')
    pkg._lib.check(lib.dcf_op_xattn(P(q.cuda()), P(k.cuda()), P(v.cuda()), P(m.cuda()), P(O), B, T, Lk, C, heads, st()))
    torch.testing.assert_close(O.cpu().view(B, T, C), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('B,T,C,heads,w', [(2, 72, 32, 4, 9), (1, 256, 256, 4, 9), (2, 64, 128, 4, 5), (1, 90, 64, 2, 19),
                                            (1, 8, 256, 4, 9), (1, 4, 128, 4, 5),
                                            # odd lengths (the second row of a wave's pair does not exist at the end of a sequence),
                                            # one-row sequences, a window wider than the unrolled variants, four chunks per row
                                            (3, 7, 256, 4, 9), (2, 1, 256, 4, 9), (2, 33, 64, 2, 19), (1, 150, 256, 4, 71),
                                            (2, 45, 1024, 16, 9)])
def test_local_attn_core(L, B, T, C, heads, w):
    pkg, lib = L
    g = torch.Generator().manual_seed(T * 3 + C)
    q, k, v = (torch.randn(B, T, C, generator=g) for _ in range(3))
    mask = torch.ones(B, T, dtype=torch.bool)
    mask[0, int(T * 0.8):] = False
    d = C // heads
    s = 1.0 / math.sqrt(math.sqrt(d))

    def split(z):
        return z.view(B, T, heads, d).permute(0, 2, 1, 3).reshape(B * heads, T, d)

    ref = R.banded_attention(split(q) * s, split(k) * s, split(v), mask, w)
    ref = ref.view(B, heads, T, d).permute(0, 2, 1, 3).reshape(B, T, C)
    O = torch.empty(B * T, C, device='cuda')
    pkg._lib.check(lib.dcf_op_local_attn(P(q.cuda()), P(k.cuda()), P(v.cuda()), P(mask.cuda()), P(O), B, T, C, heads, w, st()))
    torch.testing.assert_close(O.cpu().view(B, T, C), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('B,T,C,heads', [(2, 200, 128, 4), (1, 1000, 256, 4), (3, 64, 256, 4), (1, 37, 128, 4)])
def test_global_attn_core(L, B, T, C, heads):
    """window 0: every query attends to every valid key of its sequence (blocks.py:339-356, :374-393) -- fp64 softmax"""
    pkg, lib = L
    g = torch.Generator().manual_seed(T * 5 + C)
    q, k, v = (torch.randn(B, T, C, generator=g) for _ in range(3))
    mask = torch.ones(B, T, dtype=torch.bool)
    mask[0, int(T * 0.7):] = False
    if T > 20:
        mask[0, 5] = False
    d = C // heads

    def split(z):
        return z.double().view(B, T, heads, d).permute(0, 2, 1, 3)          # (B, h, T, d)

    att = (split(q) / d ** 0.25) @ (split(k) / d ** 0.25).transpose(2, 3)
    att = att.masked_fill(~mask[:, None, None, :], float('-inf')).softmax(-1)
    ref = (att @ split(v)).permute(0, 2, 1, 3).reshape(B, T, C)
    O = torch.full((B * T + 1, C), float('nan'), device='cuda')
    pkg._lib.check(lib.dcf_op_local_attn(P(q.cuda()), P(k.cuda()), P(v.cuda()), P(mask.cuda()), P(O), B, T, C, heads, 0, st()))
    Oc = O.cpu()
    assert torch.isnan(Oc[B * T]).all()
    torch.testing.assert_close(Oc[:B * T].view(B, T, C).double(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('D,T,nq,norm', [(256, 256, 3, 1), (1024, 4096, 1, 1), (64, 250, 11, 0), (1024, 16384, 8, 1)])
def test_sidekick(L, D, T, nq, norm):
    pkg, lib = L
    g = torch.Generator().manual_seed(D + T)
    sh, cls = torch.randn(1, D, T, generator=g), torch.randn(nq, D, generator=g)
    ref = R.sidekick_scores(sh.double(), cls.double(), bool(norm))
    out = torch.empty(nq, T, device='cuda')
    pkg._lib.check(lib.dcf_op_sidekick(P(sh.cuda()), P(cls.cuda()), P(out), D, T, nq, norm, st()))
    torch.testing.assert_close(out.cpu().double(), ref, rtol=2e-5, atol=2e-6)


def test_gate_golden(L):
    pkg, lib = L
    g = Golden('gate.npz')
    for i, c in enumerate(g.js('cases')):
        T, vl = c['T'], c['vid_len']
        correl = g.t(f'c{i}/correl').cuda().contiguous()
        mask = (torch.arange(T) < vl).cuda()
        gate = torch.empty(1, T, device='cuda')
        mo = torch.empty(1, T, dtype=torch.bool, device='cuda')
        pkg._lib.check(lib.dcf_op_gate(P(correl), P(mask), P(gate), P(mo), T, 1, c['sn'], float(c['sratio']), 1, st()))
        assert torch.equal(gate[0].cpu().to(torch.uint8), g.t(f'c{i}/gate')), c
        assert torch.equal(mo[0].cpu(), mask.cpu())
        pkg._lib.check(lib.dcf_op_gate(P(correl), P(mask), P(gate), P(mo), T, 1, c['sn'], float(c['sratio']), 0, st()))
        assert torch.equal(mo[0].cpu(), mask.cpu() & (g.t(f'c{i}/gate') != 0)), c


# ---------------------------------------------------------------------------------------------- reference operator fixtures
# (tests/golden/ops.npz / ops64.npz, generated from the reference's own blocks) through the kernels of the forward
GEMM_MODES = [('f16x3', 16), ('bf16x6', 6), ('fp32', 1)]


def scratch_model(pkg, lib, weights, prefix, **cfg):
    """dcf_model_create + dcf_model_bind of one block's parameters (never finalized)"""
    c = pkg._lib.DcfConfig()
    base = dict(D=32, E=32, TE=32, vid_heads=4, fusion_heads=4, fusion_layers=0, n_embd_convs=0, n_stem=0, n_levels=1, win=9,
                head_layers=0, sn=60, sratio=0.3, msf=1, norm=1, max_batch=8)
    base.update(cfg)
    for k, v in base.items():
        setattr(c, k, v)
    h = ctypes.c_void_p()
    pkg._lib.check(lib.dcf_model_create(ctypes.byref(c), ctypes.byref(h)), 'dcf_model_create')
    for k, v in weights.items():
        t = v.contiguous().cuda()
        shape = (ctypes.c_int64 * max(t.dim(), 1))(*(t.shape if t.dim() else (1,)))
        pkg._lib.check(lib.dcf_model_bind(h, f'{prefix}.{k}'.encode(), P(t), shape, max(t.dim(), 1)), 'dcf_model_bind')
    return h


@pytest.mark.parametrize('mode,gm', GEMM_MODES)
@pytest.mark.parametrize('stride', [1, 2])
def test_encoder_block_golden(L, stride, mode, gm):
    """TransformerEncoder.forward (blocks.py:578-591) stride 1 / 2, window 9: reference fixture enc_s{1,2}"""
    pkg, lib = L
    ops = Golden('ops.npz')
    x, mask = ops.t('x'), ops.t('mask')
    bs, E, T = x.shape
    h = scratch_model(pkg, lib, ops.sub(f'enc_s{stride}/w/'), 'e', E=E, win=9, vid_heads=4, gemm_mode=gm)
    To = T // stride
    Y = torch.empty(bs * To, E, device='cuda')
    mo = torch.empty(bs * To, dtype=torch.bool, device='cuda')
    pkg._lib.check(lib.dcf_op_encoder(h, b'e', P(tok(x)), P(mask.reshape(-1).contiguous().cuda()), bs, T, stride, P(Y), P(mo), st()),
                   'dcf_op_encoder')
    want_m = ops.t(f'enc_s{stride}/ymask')
    assert torch.equal(mo.cpu().view(bs, 1, To), want_m)
    torch.testing.assert_close(untok(Y, bs, To), ops.t(f'enc_s{stride}/y'), rtol=2e-5, atol=2e-5)
    lib.dcf_model_destroy(h)


def test_max_pool_and_dwconv_golden(L):
    """masked_max_pool1d (blocks.py:31-47) as the stride-2 encoder front end computes it: reference fixture maxpool/y; the
    three depthwise branches against the oracle (LN -> masked depthwise k3 stride 2 -> LN, blocks.py:462-470)"""
    pkg, lib = L
    ops = Golden('ops.npz')
    x, mask = ops.t('x'), ops.t('mask')
    bs, E, T = x.shape
    w = ops.sub('enc_s2/w/')
    h = scratch_model(pkg, lib, w, 'e', E=E)
    To = T // 2
    Q, K, V, S = (torch.empty(bs * To, E, device='cuda') for _ in range(4))
    xm = x * mask                                                         # the encoder masks its input first (blocks.py:581)
    pkg._lib.check(lib.dcf_op_enc_pre(h, b'e', P(tok(xm)), P(mask.reshape(-1).contiguous().cuda()), bs, T, 2, P(Q), P(K), P(V), P(S), st()),
                   'dcf_op_enc_pre')
    # the encoder multiplies the skip by ITS output mask, the nearest-downsampled one (blocks.py:586, :101-105), which is
    # what the kernel applies; masked_max_pool1d alone re-masks with the max-pooled mask: compare where the encoder keeps it
    keep = mask[:, :, ::2].float()
    torch.testing.assert_close(untok(S, bs, To), ops.t('maxpool/y') * keep, rtol=0, atol=0)
    xn = R.channel_layer_norm(xm, w['ln_attn.weight'], w['ln_attn.bias'])
    for buf, n in ((Q, 'q'), (K, 'k'), (V, 'v')):
        y, _ = R.masked_conv1d(xn, mask, w[f'attn.{n}_conv.conv.weight'], None, 2, 1, E)
        y = R.channel_layer_norm(y, w[f'attn.{n}_norm.weight'], w[f'attn.{n}_norm.bias'])
        torch.testing.assert_close(untok(buf, bs, To), y, rtol=1e-5, atol=1e-5)
    lib.dcf_model_destroy(h)


@pytest.mark.parametrize('mode,gm', GEMM_MODES)
def test_decoder_block_golden(L, mode, gm):
    """TransformerDecoder.forward (blocks.py:632-650): reference fixture ops64.npz dec (64-wide text stream, 33 keys, the
    second sequence's keys partially masked)"""
    pkg, lib = L
    o = Golden('ops64.npz')
    x, mask, kv, kvm = o.t('x'), o.t('mask'), o.t('kv'), o.t('kv_mask')
    bs, E, T = x.shape
    h = scratch_model(pkg, lib, o.sub('dec/w/'), 'd', E=E, TE=kv.size(1), fusion_heads=4, gemm_mode=gm)
    X = tok(x)
    texts = [kv[b].contiguous().cuda() for b in range(bs)]
    tmasks = [kvm[b, 0].contiguous().cuda() for b in range(bs)]
    tp = (ctypes.c_void_p * bs)(*[t.data_ptr() for t in texts])
    mp = (ctypes.c_void_p * bs)(*[t.data_ptr() for t in tmasks])
    ln = (ctypes.c_int32 * bs)(*[kv.size(2)] * bs)
    pkg._lib.check(lib.dcf_op_decoder(h, b'd', P(X), P(mask.reshape(-1).contiguous().cuda()), bs, T, tp, mp, ln, st()), 'dcf_op_decoder')
    torch.testing.assert_close(untok(X, bs, T), o.t('dec/y'), rtol=2e-5, atol=2e-5)
    lib.dcf_model_destroy(h)


def _dec_shapes(E, TE):
    return {'ln_xattn_q.weight': (E, 1), 'ln_xattn_q.bias': (E, 1), 'ln_xattn_kv.weight': (TE, 1), 'ln_xattn_kv.bias': (TE, 1),
            'xattn.q_conv.conv.weight': (E, 1, 3), 'xattn.q_norm.weight': (E, 1), 'xattn.q_norm.bias': (E, 1),
            'xattn.xattn.query.weight': (E, E, 1), 'xattn.xattn.query.bias': (E,), 'xattn.xattn.key.weight': (E, TE, 1),
            'xattn.xattn.key.bias': (E,), 'xattn.xattn.value.weight': (E, TE, 1), 'xattn.xattn.value.bias': (E,),
            'xattn.xattn.proj.weight': (2 * E, E, 1), 'xattn.xattn.proj.bias': (2 * E,), 'ln_ffn.weight': (E, 1), 'ln_ffn.bias': (E, 1),
            'ffn.fc.weight': (4 * E, E, 1), 'ffn.fc.bias': (4 * E,), 'ffn.proj.weight': (E, 4 * E, 1), 'ffn.proj.bias': (E,),
            'drop_path_ffn.scale': (1, E, 1)}


def _run_decoder(pkg, lib, sd, x, mask, kv, kvm, affine, chain_rows):
    """dcf_op_decoder on a scratch model; chain_rows = the dec_chain_min_rows option (0: the one-kernel attention half)"""
    bs, E, T = x.shape
    pkg._lib.check(lib.dcf_debug_set_option(b'dec_chain_min_rows', chain_rows))
    try:
        h = scratch_model(pkg, lib, sd, 'd', E=E, TE=kv.size(1), fusion_heads=4, gemm_mode=16, xattn_affine=int(affine))
        X = tok(x)
        texts = [kv[b, :, :int(kvm[b, 0].numel())].contiguous().cuda() for b in range(bs)]
        tmasks = [kvm[b, 0].contiguous().cuda() for b in range(bs)]
        tp = (ctypes.c_void_p * bs)(*[t.data_ptr() for t in texts])
        mp = (ctypes.c_void_p * bs)(*[t.data_ptr() for t in tmasks])
        ln = (ctypes.c_int32 * bs)(*[kv.size(2)] * bs)
        pkg._lib.check(lib.dcf_op_decoder(h, b'd', P(X), P(mask.reshape(-1).contiguous().cuda()), bs, T, tp, mp, ln, st()), 'dcf_op_decoder')
        y = untok(X, bs, T)
        lib.dcf_model_destroy(h)
        return y
    finally:
        pkg._lib.check(lib.dcf_debug_set_option(b'dec_chain_min_rows', -1))


# (bs, T, Lk, affine): one and two 32-key tiles, windows with a partial tail (T % 128 != 0), sequences whose tail is padded, a
# hole inside the valid part, partially masked keys; T = 2560 x 2 is large enough for the row statistics to ride into ffn.fc
@pytest.mark.parametrize('bs,T,Lk,affine', [(2, 200, 33, 0), (3, 128, 20, 0), (1, 450, 64, 1), (2, 2560, 33, 0)])
def test_dec_chain_vs_fp64(L, bs, T, Lk, affine):
    """the attention half of a fusion layer as one kernel (csrc/dec_chain.hip) inside dcf_op_decoder: TransformerDecoder.forward
    (blocks.py:632-650) at E = 256 against the oracle run in fp64, and against the launches the kernel replaces"""
    pkg, lib = L
    E, TE = 256, 256
    sd = pkg.synth.make_state_dict(_dec_shapes(E, TE), 4000 + T + Lk)
    g = torch.Generator().manual_seed(bs * 100 + T + Lk)
    x = torch.randn(bs, E, T, generator=g) * 1.5 + 0.3
    mask = torch.ones(bs, 1, T, dtype=torch.bool)
    for b in range(bs):
        n = int(torch.randint(T // 2, T + 1, (1,), generator=g)) if b else T
        mask[b, :, n:] = False
        if T > 140:
            mask[b, :, 127:129] = False                           # a hole across a window boundary
    kv = torch.randn(bs, TE, Lk, generator=g)
    kvm = torch.ones(bs, 1, Lk, dtype=torch.bool)
    if bs > 1:
        kvm[1, :, Lk * 2 // 3:] = False
    sdd = {'d.' + k: v.double() for k, v in sd.items()}
    want, _ = R.transformer_decoder(sdd, 'd', x.double(), mask, kv.double(), kvm, 4, adaln=not affine)
    got = _run_decoder(pkg, lib, sd, x, mask, kv, kvm, affine, 0)
    old = _run_decoder(pkg, lib, sd, x, mask, kv, kvm, affine, 1 << 30)
    torch.testing.assert_close(old.double(), want, rtol=2e-5, atol=2e-5)      # (padded rows included: they carry the shift)
    torch.testing.assert_close(got.double(), want, rtol=2e-5, atol=2e-5)


def _ops256_weights(pkg, o, block):
    meta = o.js('meta')[block]
    return pkg.synth.make_state_dict({k: tuple(v) for k, v in meta['shapes'].items()}, meta['seed'])


@pytest.mark.parametrize('chain_rows', [0, 1 << 30])
def test_decoder_block_golden_probe_width(L, chain_rows):
    """TransformerDecoder.forward (blocks.py:632-650) at E = 256, four heads, 33 keys: reference fixture ops256.npz through
    dcf_op_decoder, with the attention half as one kernel (dec_chain.hip) and as the separate launches"""
    pkg, lib = L
    o = Golden('ops256.npz')
    sd = _ops256_weights(pkg, o, 'dec')
    got = _run_decoder(pkg, lib, sd, o.t('x'), o.t('mask'), o.t('kv'), o.t('kv_mask'), 0, chain_rows)
    torch.testing.assert_close(got, o.t('dec/y'), rtol=2e-5, atol=2e-5)


def _run_encoder(pkg, lib, sd, x, mask, stride, chain_rows, win=9):
    """dcf_op_encoder on a scratch model; chain_rows = the enc_chain_min_rows option (0: the one-kernel q / k / v front half)"""
    bs, E, T = x.shape
    pkg._lib.check(lib.dcf_debug_set_option(b'enc_chain_min_rows', chain_rows))
    pkg._lib.check(lib.dcf_debug_set_option(b'enc_attn_min_rows', chain_rows))
    try:
        h = scratch_model(pkg, lib, sd, 'e', E=E, win=win, vid_heads=4, gemm_mode=16)
        To = T // stride
        Y = torch.empty(bs * To, E, device='cuda')
        mo = torch.empty(bs * To, dtype=torch.bool, device='cuda')
        pkg._lib.check(lib.dcf_op_encoder(h, b'e', P(tok(x)), P(mask.reshape(-1).contiguous().cuda()), bs, T, stride, P(Y), P(mo), st()),
                       'dcf_op_encoder')
        y = untok(Y, bs, To)
        lib.dcf_model_destroy(h)
        return y, mo.cpu().view(bs, 1, To)
    finally:
        pkg._lib.check(lib.dcf_debug_set_option(b'enc_chain_min_rows', -1))
        pkg._lib.check(lib.dcf_debug_set_option(b'enc_attn_min_rows', -1))


@pytest.mark.parametrize('chain_rows', [0, 1 << 30])
@pytest.mark.parametrize('stride', [1, 2])
def test_encoder_block_golden_probe_width(L, stride, chain_rows):
    """TransformerEncoder.forward (blocks.py:578-591) at E = 256, four heads, window 9, stride 1 / 2: reference fixture ops256.npz
    through dcf_op_encoder, with the q / k / v front half as one kernel (enc_chain.hip) and as the separate launches"""
    pkg, lib = L
    o = Golden('ops256.npz')
    sd = _ops256_weights(pkg, o, f'enc_s{stride}')
    got, gm = _run_encoder(pkg, lib, sd, o.t('x'), o.t('mask'), stride, chain_rows)
    assert torch.equal(gm, o.t(f'enc_s{stride}/ymask'))
    torch.testing.assert_close(got, o.t(f'enc_s{stride}/y'), rtol=2e-5, atol=2e-5)


def _enc_shapes(E):
    sh = {'ln_attn.weight': (E, 1), 'ln_attn.bias': (E, 1), 'ln_ffn.weight': (E, 1), 'ln_ffn.bias': (E, 1),
          'drop_path_attn.scale': (1, E, 1), 'drop_path_ffn.scale': (1, E, 1),
          'ffn.fc.weight': (4 * E, E, 1), 'ffn.fc.bias': (4 * E,), 'ffn.proj.weight': (E, 4 * E, 1), 'ffn.proj.bias': (E,)}
    for n in 'qkv':
        sh[f'attn.{n}_conv.conv.weight'] = (E, 1, 3)
        sh[f'attn.{n}_norm.weight'] = (E, 1)
        sh[f'attn.{n}_norm.bias'] = (E, 1)
    for n in ('query', 'key', 'value', 'proj'):
        sh[f'attn.attn.{n}.weight'] = (E, E, 1)
        sh[f'attn.attn.{n}.bias'] = (E,)
    return sh


# (bs, T, stride): windows with a partial tail, sequences with a padded tail, a hole across a 128-row window boundary
@pytest.mark.parametrize('bs,T,stride', [(2, 200, 1), (3, 128, 1), (1, 452, 1), (2, 2560, 1), (2, 400, 2), (2, 5120, 2)])
def test_enc_chain_vs_fp64(L, bs, T, stride):
    """the q / k / v front half of an encoder layer as one kernel (csrc/enc_chain.hip) inside dcf_op_encoder: TransformerEncoder.forward
    (blocks.py:578-591) at E = 256 against the oracle run in fp64, and against the launches the kernel replaces"""
    pkg, lib = L
    E = 256
    sd = pkg.synth.make_state_dict(_enc_shapes(E), 5000 + T)
    g = torch.Generator().manual_seed(bs * 77 + T)
    x = torch.randn(bs, E, T, generator=g) * 1.5 + 0.3
    mask = torch.ones(bs, 1, T, dtype=torch.bool)
    for b in range(bs):
        n = int(torch.randint(T // 2, T + 1, (1,), generator=g)) if b else T
        mask[b, :, n:] = False
        if T > 140:
            mask[b, :, 127:129] = False
    sdd = {'e.' + k: v.double() for k, v in sd.items()}
    want, wm = R.transformer_encoder(sdd, 'e', x.double(), mask, stride, 4, 9)
    got, gm = _run_encoder(pkg, lib, sd, x, mask, stride, 0)
    old, om = _run_encoder(pkg, lib, sd, x, mask, stride, 1 << 30)
    assert torch.equal(gm, wm) and torch.equal(om, wm)
    torch.testing.assert_close(old.double(), want, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(got.double(), want, rtol=2e-5, atol=2e-5)


def test_tcn_golden(L):
    """TCN.forward (tcn.py:66-84), 4 dilated residual layers: reference fixture tcn/y"""
    pkg, lib = L
    ops = Golden('ops.npz')
    x, mask = ops.t('tcn/x'), ops.t('mask')
    bs, n_in, T = x.shape
    h = scratch_model(pkg, lib, ops.sub('tcn/w/'), 'r')
    outs = []
    try:
        for frag in (1, 0):       # the layers' weight fragments from the per-model image / built by every workgroup: the same bits
            pkg._lib.check(lib.dcf_debug_set_option(b'tcn_frag', frag))
            Y = torch.empty(bs * T, 32, device='cuda')
            pkg._lib.check(lib.dcf_op_tcn(h, b'r', P(tok(x)), P(mask.reshape(-1).contiguous().cuda()), bs, T, n_in, 4, P(Y), st()), 'dcf_op_tcn')
            torch.testing.assert_close(untok(Y, bs, T), ops.t('tcn/y'), rtol=1e-5, atol=1e-5)
            outs.append(Y.clone())
    finally:
        pkg._lib.check(lib.dcf_debug_set_option(b'tcn_frag', -1))
    assert torch.equal(outs[0], outs[1])
    # the leading layers as ONE launch over LDS windows (k_tcn_stack; the fixture has 4 layers: 3 are stacked, the last carries conv_out)
    # against the layer-by-layer launches: the same arithmetic, the same bits -- on the fixture and on longer, ragged sequences
    # (several windows per sequence, a sequence end inside a window, masked rows)
    try:
        pkg._lib.check(lib.dcf_debug_set_option(b'tcn_stack', 0))
        Y0 = torch.empty(bs * T, 32, device='cuda')
        pkg._lib.check(lib.dcf_op_tcn(h, b'r', P(tok(x)), P(mask.reshape(-1).contiguous().cuda()), bs, T, n_in, 4, P(Y0), st()), 'dcf_op_tcn')
        assert torch.equal(Y0, outs[0])
        g = torch.Generator().manual_seed(4)
        for bs2, T2 in ((3, 1000), (2, 4096), (1, 37)):
            x2 = torch.randn(bs2 * T2, n_in, generator=g).cuda()
            m2 = (torch.rand(bs2 * T2, generator=g) > 0.15).to(torch.uint8).cuda()
            got = []
            for nl in (0, 2, 3):
                pkg._lib.check(lib.dcf_debug_set_option(b'tcn_stack', nl))
                Y2 = torch.empty(bs2 * T2, 32, device='cuda')
                pkg._lib.check(lib.dcf_op_tcn(h, b'r', P(x2), P(m2), bs2, T2, n_in, 4, P(Y2), st()), 'dcf_op_tcn')
                got.append(Y2.clone())
            assert torch.equal(got[0], got[1]) and torch.equal(got[0], got[2]), (bs2, T2)
    finally:
        pkg._lib.check(lib.dcf_debug_set_option(b'tcn_stack', -1))
    lib.dcf_model_destroy(h)


def _linear(pkg, lib, x_rows, w, b, nterms):
    M, K = x_rows.shape
    N = w.size(0)
    C = torch.empty(M, N, device='cuda')
    wv = w.reshape(N, K).contiguous().cuda()
    if nterms:
        pkg._lib.check(lib.dcf_op_linear_split(P(x_rows), P(wv), P(b.cuda()), P(C), M, N, K, 0, nterms, st()))
    else:
        pkg._lib.check(lib.dcf_op_linear(P(x_rows), P(wv), P(b.cuda()), P(C), M, N, K, 0, st()))
    return C


@pytest.mark.parametrize('nterms', [16, 6, 0])
def test_mha_global_golden(L, nterms):
    """MaskedMHA global branch incl. its four 1x1 projections (blocks.py:348-356,374-393): reference fixture ops64.npz
    mha_global (kv_dim 64, out_dim 2E), core = dcf_op_xattn"""
    pkg, lib = L
    o = Golden('ops64.npz')
    x, kv, kvm = o.t('x'), o.t('kv'), o.t('kv_mask')
    w = o.sub('mha_global/w/')
    bs, E, T = x.shape
    Lk = kv.size(2)
    q = _linear(pkg, lib, tok(x), w['query.weight'], w['query.bias'], nterms)
    k = _linear(pkg, lib, tok(kv), w['key.weight'], w['key.bias'], nterms)
    v = _linear(pkg, lib, tok(kv), w['value.weight'], w['value.bias'], nterms)
    O = torch.empty(bs * T, E, device='cuda')
    pkg._lib.check(lib.dcf_op_xattn(P(q), P(k), P(v), P(kvm.reshape(-1).contiguous().cuda()), P(O), bs, T, Lk, E, 4, st()))
    y = _linear(pkg, lib, O, w['proj.weight'], w['proj.bias'], nterms)
    torch.testing.assert_close(untok(y, bs, T), o.t('mha_global/y'), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('win', [5, 9, 19])
def test_mha_local_golden(L, win):
    """MaskedMHA local branch incl. projections (blocks.py:357-373,391-392): reference fixtures mha_local{5,9,19}"""
    pkg, lib = L
    ops = Golden('ops.npz')
    x, mask = ops.t('x'), ops.t('mask')
    w = ops.sub(f'mha_local{win}/w/')
    bs, E, T = x.shape
    xr = tok(x)
    q, k, v = (_linear(pkg, lib, xr, w[f'{n}.weight'], w[f'{n}.bias'], 16) for n in ('query', 'key', 'value'))
    O = torch.empty(bs * T, E, device='cuda')
    pkg._lib.check(lib.dcf_op_local_attn(P(q), P(k), P(v), P(mask.reshape(-1).contiguous().cuda()), P(O), bs, T, E, 4, win, st()))
    y = _linear(pkg, lib, O, w['proj.weight'], w['proj.bias'], 16)
    torch.testing.assert_close(untok(y, bs, T), ops.t(f'mha_local{win}/y'), rtol=1e-5, atol=1e-5)


def test_xattn_core_config2_shape(L):
    """BASELINE configs[1] as bench.py measures it -- 8 queries x T = 4096 clips, E = 1024, 16 heads, 33 keys (one query
    with half of its keys padded): the grid-stride instantiation of k_xattn_mfma against the oracle's MaskedMHA global
    core (identity projections, oracle/decafnet_ref.py _mha_global_qkv = blocks.py:374-389)"""
    pkg, lib = L
    B, T, Lk, C, heads = 8, 4096, 33, 1024, 16
    g = torch.Generator().manual_seed(4096)
    q, k, v = torch.randn(B, T, C, generator=g), torch.randn(B, Lk, C, generator=g), torch.randn(B, Lk, C, generator=g)
    m = torch.ones(B, Lk, dtype=torch.bool)
    m[-1, Lk // 2:] = False
    eye = torch.eye(C)[:, :, None]
    sd = {f'a.{n}.weight': eye for n in ('query', 'key', 'value', 'proj')}
    want = R._mha_global_qkv(sd, 'a', q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), m[:, None, :], heads).transpose(1, 2)
    O = torch.empty(B * T, C, device='cuda')
    pkg._lib.check(lib.dcf_op_xattn(P(q.cuda()), P(k.cuda()), P(v.cuda()), P(m.cuda()), P(O), B, T, Lk, C, heads, st()))
    torch.testing.assert_close(O.cpu().view(B, T, C), want, rtol=1e-5, atol=1e-5)


# ---------------------------------------------------------------------------------------------- NMS
def test_nms_known_answers(L):
    pkg, lib = L
    nms = pkg.nms
    g = Golden('nms_kat.npz')
    for i, c in enumerate(g.js('cases')):
        segs, scores = g.t(f'k{i}/segs'), g.t(f'k{i}/scores')
        assert torch.equal(nms.nms(segs, scores, c['iou_thresh']), g.t(f'k{i}/nms')), c
        for method in (0, 1, 2):
            dets = torch.full((len(segs), 3), -7.0)
            idx = nms.softnms(segs, scores, dets, c['iou_thresh'], c['sigma'], c['min_score'], method)
            assert torch.equal(idx, g.t(f'k{i}/soft{method}/idx')), (c, method)
            want = g.t(f'k{i}/soft{method}/dets')
            # bit for bit since round 5: the Gaussian weight is the reference C library's expf restated (postproc.hip expf_glibc)
            assert torch.equal(dets[:len(idx)], want), (c, method, int((dets[:len(idx)] != want).sum()))
            if len(idx) < len(segs):
                assert (dets[len(idx):] == -7.0).all()


def test_nms_known_answers_beyond_4096_candidates(L):
    """n = 4097 / 6000 / 8192: more candidates than one workgroup's LDS holds -- the same kernels over a global scratch block;
    expected indices / dets from the reference's extension (the reference takes any n, nms_cpu.cpp:20-63)"""
    pkg, lib = L
    nms = pkg.nms
    g = Golden('nms_kat_big.npz')
    for i, c in enumerate(g.js('cases')):
        segs, scores = g.t(f'k{i}/segs'), g.t(f'k{i}/scores')
        assert torch.equal(nms.nms(segs, scores, c['iou_thresh']), g.t(f'k{i}/nms')), c
        for method in (1, 2):
            dets = torch.full((len(segs), 3), -7.0)
            idx = nms.softnms(segs, scores, dets, c['iou_thresh'], c['sigma'], c['min_score'], method)
            assert torch.equal(idx, g.t(f'k{i}/soft{method}/idx')), (c, method)
            assert torch.equal(dets[:len(idx)], g.t(f'k{i}/soft{method}/dets')), (c, method)
    # batched_nms over device tensors of that size (both modes), against the oracle's composition
    segs, scores = g.t('k1/segs'), g.t('k1/scores')
    for mode in ('nms', 'soft_nms'):
        s, c = nms.batched_nms(segs.cuda(), scores.cuda(), 0.5, 0.0 if mode == 'nms' else 0.001, 50, mode=mode, sigma=0.9, voting_thresh=0.0)
        if mode == 'nms':
            keep = nms_oracle.nms(segs, scores, 0.5)[:50]
            assert torch.equal(s.cpu(), segs[keep]) and torch.equal(c.cpu(), scores[keep])
        else:
            d = torch.zeros(len(segs), 3)
            nms_oracle.softnms(segs, scores, d, 0.5, 0.9, 0.001, 2)
            assert torch.equal(c.cpu(), d[:50, 2])


def test_nms_module_abi(L):
    """the drop-in module: name, keyword names, CPU/contiguity/dtype checks (nms_cpu.cpp:11-17,184-194)"""
    import nms_1d_cpu_vg as ext
    segs = torch.tensor([[0.0, 10.0], [1.0, 11.0], [20.0, 30.0]])
    scores = torch.tensor([0.3, 0.9, 0.5])
    idx = ext.nms(segs=segs, scores=scores, iou_thresh=0.5)
    assert idx.dtype == torch.int64 and idx.tolist() == [1, 2]
    dets = torch.zeros(3, 3)
    idx = ext.softnms(segs=segs, scores=scores, dets=dets, iou_thresh=0.5, sigma=0.5, min_score=0.001, method=0)
    assert idx.tolist() == [1, 2] and dets[0].tolist() == [1.0, 11.0, pytest.approx(0.9)]
    assert ext.nms(torch.zeros(0, 2), torch.zeros(0), 0.5).shape == (0,)
    with pytest.raises(RuntimeError, match='must be a CPU tensor'):
        ext.nms(segs.cuda(), scores, 0.5)
    with pytest.raises(RuntimeError, match='must be contiguous'):
        ext.nms(torch.zeros(3, 4)[:, :2], scores, 0.5)
    with pytest.raises(RuntimeError, match='expected scalar type Float'):
        ext.nms(segs.double(), scores, 0.5)


def test_nms_fuzz_vs_oracle(L):
    pkg, lib = L
    nms = pkg.nms
    g = torch.Generator().manual_seed(321)
    for trial in range(28):
        # 25 random sizes, then the default pre-NMS top-k (2000), the LDS capacity (4096) and one past it (global scratch)
        n = int(torch.randint(1, 1500, (1,), generator=g)) if trial < 25 else (2000, 4096, 4097)[trial - 25]
        c = torch.rand(n, generator=g) * (20 + 3 * n ** 0.5)
        ln = torch.rand(n, generator=g) * 30 + 0.1
        segs = torch.stack((c - ln / 2, c + ln / 2), -1).contiguous()
        scores = torch.rand(n, generator=g).contiguous()
        thr = float(torch.rand(1, generator=g) * 0.8 + 0.05)
        assert torch.equal(nms.nms(segs, scores, thr), nms_oracle.nms(segs, scores, thr)), (trial, n)
        for method in (0, 1, 2):
            d1, d2 = torch.zeros(n, 3), torch.zeros(n, 3)
            ms = float(torch.rand(1, generator=g) * 0.2)
            i1 = nms.softnms(segs, scores, d1, thr, 0.5, ms, method)
            i2 = nms_oracle.softnms(segs, scores, d2, thr, 0.5, ms, method)
            assert torch.equal(i1, i2), (trial, n, method)
            assert torch.equal(d1[:len(i1)], d2[:len(i2)]), (trial, n, method)       # (the oracle's expf is the host C library's)


def test_softnms_ties_and_pruning_fuzz_vs_oracle(L):
    """Quantised scores and segments: decayed scores tie again and again, and fall below min_score one at a time or several at once --
    the order of the picks then hangs on the positions the reference's swap-with-last pruning (nms_cpu.cpp:157-165) leaves behind, which
    both pruning paths of k_softnms (one dead segment: the move alone; several: the scans) have to reproduce."""
    pkg, lib = L
    nms = pkg.nms
    g = torch.Generator().manual_seed(99)
    for trial in range(36):
        n = int(torch.randint(2, 900, (1,), generator=g)) if trial < 33 else (2000, 4096, 4100)[trial - 33]
        levels = (4, 16, 64)[trial % 3]
        c = torch.randint(0, 12 + n // 8, (n,), generator=g).float() * 2.0
        ln = torch.randint(1, 6, (n,), generator=g).float() * 2.0
        segs = torch.stack((c - ln / 2, c + ln / 2), -1).contiguous()
        scores = (torch.randint(1, levels + 1, (n,), generator=g).float() / levels).contiguous()
        thr = (0.1, 0.3, 0.5)[trial % 3]
        for method in (0, 1, 2):
            for ms in (0.001, 0.2, 0.45):
                d1, d2 = torch.zeros(n, 3), torch.zeros(n, 3)
                i1 = nms.softnms(segs, scores, d1, thr, 0.5, ms, method)
                i2 = nms_oracle.softnms(segs, scores, d2, thr, 0.5, ms, method)
                assert torch.equal(i1, i2), (trial, n, method, ms)
                assert torch.equal(d1[:len(i1)], d2[:len(i2)]), (trial, n, method, ms)


def test_nms_ties_are_stable(L):
    """all scores equal: our definition = lowest index first (documented deviation from at::sort)"""
    pkg, _ = L
    segs = torch.tensor([[0.0, 1.0], [10.0, 11.0], [20.0, 21.0], [0.2, 1.2]])
    scores = torch.full((4,), 0.5)
    assert pkg.nms.nms(segs, scores, 0.5).tolist() == [0, 1, 2]
    assert torch.equal(pkg.nms.nms(segs, scores, 0.5), nms_oracle.nms(segs, scores, 0.5))


def test_collect_and_batched_nms_golden(L):
    pkg, lib = L
    g = Golden('postproc.npz')
    meta = g.js('meta')
    Lv, T0 = meta['L'], meta['T0']
    logits = torch.cat([g.t(f'l{l}/logits')[0] for l in range(Lv)])[None].cuda()
    offsets = torch.cat([g.t(f'l{l}/offsets')[0] for l in range(Lv)])[None].cuda()
    masks = torch.cat([g.t(f'l{l}/mask')[0] for l in range(Lv)])[None].cuda()
    segs, scores, counts = pkg.nms.collect_segments(logits, offsets, masks, T0, Lv)
    n = int(counts[0])
    want_segs, want_scores = g.t('segs'), g.t('scores')
    assert n == len(want_scores)
    torch.testing.assert_close(scores[0, :n].cpu(), want_scores, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(segs[0, :n].cpu(), want_segs, rtol=1e-6, atol=1e-5)
    for k, cfg in g.js('nms_cfgs').items():
        s, c = pkg.nms.batched_nms(want_segs.clone(), want_scores.clone(), **cfg)
        torch.testing.assert_close(c, g.t(f'{k}/scores'), rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(s, g.t(f'{k}/segs'), rtol=1e-5, atol=1e-4)
        s2, c2 = pkg.nms.batched_nms(want_segs.cuda(), want_scores.cuda(), **cfg)
        assert s2.is_cuda and torch.equal(s2.cpu(), s)


def test_batched_nms_over_queries_without_host_sync(L):
    """batched_nms_queries (all queries of a video, device only) == batched_nms query by query == the reference fixture"""
    pkg, lib = L
    g = Golden('postproc.npz')
    want_segs, want_scores = g.t('segs'), g.t('scores')
    n = len(want_scores)
    for k, cfg in g.js('nms_cfgs').items():
        s, c, kc = pkg.nms.batched_nms_queries(want_segs[None].cuda(), want_scores[None].cuda(), torch.tensor([n], dtype=torch.int32).cuda(), **cfg)
        kk = int(kc[0])
        assert kk == len(g.t(f'{k}/scores')), (k, kk)
        torch.testing.assert_close(c[0, :kk].cpu(), g.t(f'{k}/scores'), rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(s[0, :kk].cpu(), g.t(f'{k}/segs'), rtol=1e-5, atol=1e-4)
    # several queries with different candidate counts (incl. none and fewer than max_num_segs), rows sorted by score
    gen = torch.Generator().manual_seed(77)
    nq, K = 5, 600
    counts = torch.tensor([600, 3, 0, 257, 1], dtype=torch.int32)
    ctr = torch.rand(nq, K, generator=gen) * 300
    ln = torch.rand(nq, K, generator=gen) * 40 + 0.2
    segs = torch.stack((ctr - ln / 2, ctr + ln / 2), -1).contiguous()
    scores = torch.rand(nq, K, generator=gen).sort(1, descending=True)[0].contiguous()
    for cfg in (dict(iou_thresh=0.1, min_score=0.001, max_num_segs=5, mode='soft_nms', sigma=0.9, voting_thresh=0.95),
                dict(iou_thresh=0.5, min_score=0.3, max_num_segs=7, mode='nms', sigma=0.5, voting_thresh=0.0),
                dict(iou_thresh=0.4, min_score=0.0, max_num_segs=4, mode='nms', sigma=0.5, voting_thresh=0.75),
                dict(iou_thresh=0.3, min_score=0.05, max_num_segs=6, mode='soft_nms', sigma=0.4, voting_thresh=0.0)):
        s, c, kc = pkg.nms.batched_nms_queries(segs.cuda(), scores.cuda(), counts.cuda(), **cfg)
        for q in range(nq):
            m = int(counts[q])
            ws, wc = pkg.nms.batched_nms(segs[q, :m].clone(), scores[q, :m].clone(), **cfg)
            kk = int(kc[q])
            assert kk == len(wc), (cfg, q, kk, len(wc))
            torch.testing.assert_close(c[q, :kk].cpu(), wc, rtol=0, atol=0)
            torch.testing.assert_close(s[q, :kk].cpu(), ws, rtol=1e-6, atol=1e-6)


def test_collect_multi_query_vs_oracle(L):
    pkg, lib = L
    g = torch.Generator().manual_seed(9)
    T0, Lv, nq = 4096, 8, 3
    S = sum(T0 >> l for l in range(Lv))
    logits = torch.randn(nq, S, generator=g) * 2 - 2
    offsets = torch.rand(nq, S, 2, generator=g) * 5
    masks = torch.ones(nq, S, dtype=torch.bool)
    pts = R.generate_points(T0, Lv, 4, 0.5)
    sizes = [T0 >> l for l in range(Lv)]
    segs, scores, counts = pkg.nms.collect_segments(logits.cuda(), offsets.cuda(), masks.cuda(), T0, Lv)
    for q in range(nq):
        ws, wc = R.collect_segments(pts, [x[None] for x in logits[q].split(sizes)], [x[None] for x in offsets[q].split(sizes)],
                                    [x[None] for x in masks[q].split(sizes)])
        n = int(counts[q])
        assert n == len(wc)
        torch.testing.assert_close(scores[q, :n].cpu(), wc, rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(segs[q, :n].cpu(), ws, rtol=1e-6, atol=1e-4)


@pytest.mark.parametrize('T0,Lv,topk,shift,quant', [
    (32768, 2, 4000, -2.0, 0),      # 49152 points: keys re-read from global memory; 4000 kept: the bitonic network, 4 keys per lane
    (16384, 8, 2000, -2.0, 0),      # the bench shape: keys in registers, radix sort
    (16384, 8, 2000, -9.0, 0),      # few candidates above the score threshold (fewer than top-k)
    (4096, 8, 300, 0.0, 16),        # logits quantised to 16 levels: many exactly equal scores, ties broken by the lower index
    (64, 1, 2000, 0.0, 0),          # fewer points than threads
])
def test_collect_paths_vs_oracle(L, T0, Lv, topk, shift, quant):
    """every code path of the proposal decoder against the oracle: scores exact, order exact (stable), segments 1e-6"""
    pkg, lib = L
    g = torch.Generator().manual_seed(T0 + topk)
    S = sum(T0 >> l for l in range(Lv))
    logits = torch.randn(1, S, generator=g) * 2 + shift
    if quant:
        logits = torch.round(logits * quant / 8) * 8 / quant
    offsets = torch.rand(1, S, 2, generator=g) * 5
    masks = torch.ones(1, S, dtype=torch.bool)
    masks[0, int(S * 0.9):] = False
    pts = R.generate_points(T0, Lv, 4, 0.5)
    sizes = [T0 >> l for l in range(Lv)]
    segs, scores, counts = pkg.nms.collect_segments(logits.cuda(), offsets.cuda(), masks.cuda(), T0, Lv, pre_nms_topk=topk)
    ws, wc = R.collect_segments(pts, [x[None] for x in logits[0].split(sizes)], [x[None] for x in offsets[0].split(sizes)],
                                [x[None] for x in masks[0].split(sizes)], pre_nms_topk=topk)
    n = int(counts[0])
    assert n == len(wc), (n, len(wc))
    torch.testing.assert_close(scores[0, :n].cpu(), wc, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(segs[0, :n].cpu(), ws, rtol=1e-6, atol=1e-4)


def test_collect_with_ext_scores(L):
    """external per-clip scores, max-pooled down the pyramid (worker_v2.py:1150-1156): reference fixture at nq = 1 and the
    oracle at nq = 3 with per-query rows"""
    pkg, lib = L
    g = Golden('postproc_ext.npz')
    meta = g.js('meta')
    Lv, T0 = meta['L'], meta['T0']
    logits = torch.cat([g.t(f'l{l}/logits')[0] for l in range(Lv)])[None].cuda()
    offsets = torch.cat([g.t(f'l{l}/offsets')[0] for l in range(Lv)])[None].cuda()
    masks = torch.cat([g.t(f'l{l}/mask')[0] for l in range(Lv)])[None].cuda()
    segs, scores, counts = pkg.nms.collect_segments(logits, offsets, masks, T0, Lv, pre_nms_topk=meta['pre_nms_topk'],
                                                    ext_scores=g.t('ext').cuda())
    n = int(counts[0])
    assert n == len(g.t('scores'))
    torch.testing.assert_close(scores[0, :n].cpu(), g.t('scores'), rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(segs[0, :n].cpu(), g.t('segs'), rtol=1e-6, atol=1e-5)

    gen = torch.Generator().manual_seed(19)
    T0, Lv, nq = 2048, 7, 3
    S = sum(T0 >> l for l in range(Lv))
    logits = torch.randn(nq, S, generator=gen) * 2 - 1
    offsets = torch.rand(nq, S, 2, generator=gen) * 5
    masks = torch.ones(nq, S, dtype=torch.bool)
    ext = torch.rand(nq, T0, generator=gen)
    ext[:, ::3] = 0
    pts = R.generate_points(T0, Lv, 4, 0.5)
    sizes = [T0 >> l for l in range(Lv)]
    segs, scores, counts = pkg.nms.collect_segments(logits.cuda(), offsets.cuda(), masks.cuda(), T0, Lv, ext_scores=ext.cuda())
    for q in range(nq):
        ws, wc = R.collect_segments(pts, [x[None] for x in logits[q].split(sizes)], [x[None] for x in offsets[q].split(sizes)],
                                    [x[None] for x in masks[q].split(sizes)], ext_scores=ext[q])
        n = int(counts[q])
        assert n == len(wc)
        torch.testing.assert_close(scores[q, :n].cpu(), wc, rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(segs[q, :n].cpu(), ws, rtol=1e-6, atol=1e-4)


@pytest.mark.gpu
def test_mfma_rate_calibration_is_sane(L):
    """dcf_calib_mfma_rate (bench.py's roofline.checks.mfma_sustained): one 32x32x16 MFMA occupies a SIMD for 32 cycles, i.e. 13.3 ns at
    the nominal 2.4 GHz and never less; under load the part holds 1.4 - 2.4 GHz.  The 16x16x32 shape is half of that."""
    import ctypes
    pkg, lib = L
    ncu, ns32, ns16 = ctypes.c_int32(0), ctypes.c_float(0.), ctypes.c_float(0.)
    pkg._lib.check(lib.dcf_calib_mfma_rate(0, 1 << 14, ctypes.byref(ncu), ctypes.byref(ns32)))
    pkg._lib.check(lib.dcf_calib_mfma_rate(1, 1 << 15, ctypes.byref(ncu), ctypes.byref(ns16)))
    assert ncu.value >= 64
    assert 32 / 2.6 <= ns32.value <= 32 / 1.0, ns32.value
    assert 16 / 2.6 <= ns16.value <= 16 / 1.0, ns16.value
    assert lib.dcf_calib_mfma_rate(2, 1 << 14, ctypes.byref(ncu), ctypes.byref(ns32)) != 0      # unknown shape: an error, not a crash
