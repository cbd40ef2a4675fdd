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
                                         (65600, 256, 256, 1), (32704, 288, 96, 0), (20000, 256, 128, 2)])   # 128x256 / 128x96 / 64x256 tiles
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


@pytest.mark.parametrize('M,K,relu,raw', [(28700, 96, 1, False), (32768, 256, 0, True)])
def test_linear_with_fused_layernorm(L, M, K, relu, raw):
    """GEMM with the channel LayerNorm (+ReLU) of the output row in its epilogue vs fp64 (last row tile partial)"""
    pkg, lib = L
    N = 256
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
                                        P(C) if raw else None, P(Y), M, N, K, relu, 6, st()))
    torch.testing.assert_close(Y.cpu().double(), ref, rtol=5e-5, atol=5e-5)
    if raw:
        torch.testing.assert_close(C.cpu().double(), v, rtol=2e-5, atol=2e-5)


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
                                             (1, 100, 33, 1024, 16), (3, 40, 6, 32, 4)])
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
                                            (1, 8, 256, 4, 9), (1, 4, 128, 4, 5)])
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
            torch.testing.assert_close(dets[:len(idx)], want, rtol=1e-6, atol=1e-7)
            if len(idx) < len(segs):
                assert (dets[len(idx):] == -7.0).all()


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
    for trial in range(25):
        n = int(torch.randint(1, 1500, (1,), generator=g))
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
            torch.testing.assert_close(d1[:len(i1)], d2[:len(i2)], rtol=1e-6, atol=1e-7)


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


@pytest.mark.parametrize('variant', [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize('M,E,res', [(64, 128, False), (1000, 256, True), (16500, 256, True), (333, 128, True)])
def test_fused_ffn(L, M, E, res, variant):
    """fc + GELU + proj in one kernel (ffn_f16.hip, f16x3) vs fp64: C = R + ls * (GELU(X W1^T + b1) W2^T + b2) * mask"""
    pkg, lib = L
    g = torch.Generator().manual_seed(M + E)
    X = torch.randn(M, E, generator=g) * 2
    W1 = torch.randn(4 * E, E, generator=g) / math.sqrt(E)
    b1 = torch.randn(4 * E, generator=g) * 0.5
    W2 = torch.randn(E, 4 * E, generator=g) / math.sqrt(4 * E)
    b2 = torch.randn(E, generator=g)
    ref = F.gelu(X.double() @ W1.double().t() + b1.double()) @ W2.double().t() + b2.double()
    R = ls = mask = None
    if res:
        R = torch.randn(M, E, generator=g)
        ls = torch.rand(E, generator=g) + 0.5
        mask = torch.rand(M, generator=g) > 0.2
        ref = R.double() + ls.double() * (ref * mask.double()[:, None])
    C = torch.empty(M, E, device='cuda')
    dev = lambda t: None if t is None else t.cuda()
    Rd, lsd, md = dev(R), dev(ls), dev(mask)
    pkg._lib.check(lib.dcf_op_ffn(P(X.cuda()), P(W1.cuda()), P(b1.cuda()), P(W2.cuda()), P(b2.cuda()), P(Rd) if res else None,
                                  P(lsd) if res else None, P(md) if res else None, P(C), M, E, variant, st()), 'dcf_op_ffn')
    torch.testing.assert_close(C.cpu().double(), ref, rtol=3e-5, atol=3e-5)
