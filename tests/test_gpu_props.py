"""Size-independent properties of the hot path at BASELINE's full sizes, where the oracle is too slow to be the checker for every case:
NMS (sortedness, separation, completeness, idempotence), soft-NMS (monotone scores, unique indices, the vanilla method against hard NMS),
the gate (block structure, kept fraction), and the forward's invariance to masked padding.  Everything goes through the C ABI."""
import math

import pytest
import torch

from conftest import load_pkg

pytestmark = pytest.mark.gpu


def rand_segments(n, span, max_len, seed, nq=1):
    g = torch.Generator().manual_seed(seed)
    c = torch.rand(nq, n, generator=g) * span
    w = torch.rand(nq, n, generator=g) * max_len + 1.0
    segs = torch.stack([c - w / 2, c + w / 2], -1).contiguous()
    scores = torch.rand(nq, n, generator=g)
    return segs, scores


def iou_1d(a, b):
    """a (n, 2), b (m, 2) -> (n, m), the reference's formula (nms_cpu.cpp:38-46: areas carry + 1e-6... only in soft-NMS; hard NMS: plain lengths)"""
    inter = (torch.minimum(a[:, None, 1], b[None, :, 1]) - torch.maximum(a[:, None, 0], b[None, :, 0])).clamp(min=0)
    la, lb = (a[:, 1] - a[:, 0])[:, None], (b[:, 1] - b[:, 0])[None, :]
    return inter / (la + lb - inter)


@pytest.mark.parametrize('n,thr', [(2000, 0.5), (4096, 0.3), (8192, 0.7), (20000, 0.5)])
def test_nms_properties_at_full_size(n, thr):
    """nms_1d_cpu semantics (nms_cpu.cpp:20-63) by their consequences: kept indices in descending score order, no two kept segments
    overlap by more than the threshold, every dropped candidate overlaps a kept one of higher (or equal, earlier) score by more than
    it, and the kept set is a fixed point."""
    pkg = load_pkg()
    segs, scores = rand_segments(n, 16000.0, 400.0, 100 + n, nq=2)
    keep, kc = pkg.nms.nms_device(segs.cuda(), scores.cuda(), None, n, n, thr)
    for q in range(2):
        k = int(kc[q])
        idx = keep[q, :k].cpu()
        assert k > 0 and len(set(idx.tolist())) == k and int(idx.min()) >= 0 and int(idx.max()) < n
        s = scores[q][idx]
        assert bool((s[:-1] >= s[1:]).all()), 'kept indices are not in descending score order'
        kept = segs[q][idx].double()
        iou = iou_1d(kept, kept)
        iou.fill_diagonal_(0)
        assert float(iou.max()) <= thr + 1e-6, 'two kept segments overlap by more than the threshold'
        dropped = torch.ones(n, dtype=torch.bool)
        dropped[idx] = False
        d_idx = dropped.nonzero().flatten()
        iou_d = iou_1d(segs[q][d_idx].double(), kept)                         # (dropped, kept)
        higher = scores[q][idx][None, :] >= scores[q][d_idx][:, None]
        assert bool(((iou_d > thr - 1e-6) & higher).any(1).all()), 'a dropped candidate has no kept suppressor'
        # fixed point: NMS of the kept set keeps all of it, in the same order
        ks, ksc = segs[q][idx][None].contiguous().cuda(), scores[q][idx][None].contiguous().cuda()
        keep2, kc2 = pkg.nms.nms_device(ks, ksc, None, k, k, thr)
        assert int(kc2[0]) == k and torch.equal(keep2[0, :k].cpu(), torch.arange(k))


@pytest.mark.parametrize('n', [2000, 4096, 6000])
def test_softnms_properties_at_full_size(n):
    """softnms_1d_cpu (nms_cpu.cpp:72-172): picks come out with non-increasing scores, every index at most once, the first pick is the
    global maximum with its score untouched; the vanilla method (weight 0 above the threshold, pruning by min_score) keeps exactly what
    hard NMS keeps, in the same order, for positive scores."""
    pkg = load_pkg()
    segs, scores = rand_segments(n, 16000.0, 400.0, 300 + n)
    scores = scores * 0.9 + 0.05
    for method in (2, 1):
        dets, inds, oc = pkg.nms.softnms_device(segs.cuda(), scores.cuda(), None, n, n, 0.3, 0.5, 0.001, method)
        k = int(oc[0])
        d, i = dets[0, :k].cpu(), inds[0, :k].cpu()
        assert k > 0 and len(set(i.tolist())) == k
        assert bool((d[:-1, 2] >= d[1:, 2]).all()), 'picked scores increase'
        assert int(i[0]) == int(scores[0].argmax()) and float(d[0, 2]) == float(scores[0].max())
        assert torch.equal(d[:, :2], segs[0][i]), 'a detection does not carry its candidate\'s segment'
        assert bool((d[:, 2] <= scores[0][i] + 1e-7).all()), 'a score grew'
    dets, inds, oc = pkg.nms.softnms_device(segs.cuda(), scores.cuda(), None, n, n, 0.4, 0.5, 0.001, 0)
    keep, kc = pkg.nms.nms_device(segs.cuda(), scores.cuda(), None, n, n, 0.4)
    # (the soft-NMS areas carry + 1e-6, nms_cpu.cpp:96, the hard ones do not: with random real-valued segments no IoU sits on the threshold)
    assert int(oc[0]) == int(kc[0]) and torch.equal(inds[0, :int(oc[0])].cpu(), keep[0, :int(kc[0])].cpu())


@pytest.mark.parametrize('T,vid_len', [(16384, 16384), (65536, 65530), (65536, 40000)])
def test_gate_properties_at_full_size(T, vid_len):
    """model.py:509-523 by its consequences: the gate is constant on the blocks nearest-neighbour interpolation assigns, the number of
    open blocks is int(sratio * n_blocks), closed beyond vid_len, and every open block's pooled score is >= every closed block's."""
    import ctypes
    pkg = load_pkg()
    lib = pkg._lib.lib()
    P = pkg._lib.ptr
    sn, ratio = 60, 0.3
    g = torch.Generator().manual_seed(T + vid_len)
    correl = torch.randn(1, T, generator=g)
    gate = torch.empty(1, T, device='cuda')
    mask = (torch.arange(T) < vid_len).cuda()
    mo = torch.empty(1, T, dtype=torch.bool, device='cuda')
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    pkg._lib.check(lib.dcf_op_gate(P(correl.cuda().contiguous()), P(mask), P(gate), P(mo), T, 1, sn, float(ratio), 1, st))
    assert torch.equal(mo[0].cpu(), mask.cpu())
    gt = gate.cpu()[0]
    assert set(gt.unique().tolist()) <= {0.0, 1.0} and float(gt[vid_len:].abs().sum()) == 0.0
    nb = (vid_len + sn - 1) // sn
    k = int(ratio * nb)
    # block of clip t under F.interpolate(mode='nearest', size=vid_len) of nb values: floor(t * nb / vid_len)
    t = torch.arange(vid_len, dtype=torch.float64)
    blk = torch.floor(t * (nb / vid_len)).long().clamp(max=nb - 1)
    per_block = torch.zeros(nb).index_add_(0, blk, gt[:vid_len]) / torch.bincount(blk, minlength=nb).clamp(min=1)
    assert set(per_block.unique().tolist()) <= {0.0, 1.0}, 'the gate changes inside a block'
    assert int(per_block.sum()) == k, (int(per_block.sum()), k)
    pooled = torch.nn.functional.avg_pool1d(correl[:, None, :vid_len], sn, sn, ceil_mode=True)[0, 0]
    if 0 < k < nb:
        assert float(pooled[per_block > 0].min()) >= float(pooled[per_block == 0].max())


def test_forward_is_invariant_to_masked_padding_at_bench_scale():
    """A video of 12 000 valid clips padded to T = 16 384 and to T = 32 768 (model.py pads to a multiple of the chunk size; the masks carry
    the length): every output at a valid position of the levels both pyramids share is the same -- MaskedConv1D, the masked attention and
    the gate (which sees vid_len only) make the padding invisible.  Probe hyper-parameters, both forwards on the chain-kernel paths."""
    pkg = load_pkg()
    L = 8
    kw = dict(D=1024, E=256, TE=256, text_in=128, n_levels=L, win=9, n_heads=4, sn=60, sratio=0.3, msf=True, norm=True,
              max_seq_len=32768, text_layers=2, text_max_len=48)      # (T <= max_seq_len: the position encoding is a prefix of one table, video_net.py:141-151)
    opt = pkg.config.make_opt(**kw)
    model = pkg.modeling.create_model(opt)
    model.load_state_dict(pkg.synth.make_state_dict({k: list(v.shape) for k, v in model.state_dict().items()}, 7))
    model = model.cuda().eval().requires_grad_(False)
    vid_len = 12000
    inp = pkg.synth.make_inputs(1024, 32768, vid_len, 1, 128, 32, 8)
    tm = [model.encode_text(t[None].cuda(), torch.ones(1, 1, t.size(-1), dtype=torch.bool, device='cuda')) for t in inp['tokens']]
    outs = []
    for T in (16384, 32768):
        got = model(inp['vid'][..., :T].contiguous().cuda(), inp['shallow_vid'][..., :T].contiguous().cuda(), inp['vid_masks'][..., :T].contiguous().cuda(),
                    tuple(t for t, _ in tm), inp['text_cls'].cuda(), tuple(m for _, m in tm), eval=True)
        outs.append([[x.cpu() for x in got[i][0]] for i in range(3)])
    assert model.numerics_status() & 1 == 0
    worst = 0.0
    for l in range(L):
        n_valid = math.ceil(vid_len / 2 ** l)
        ma, mb = outs[0][2][l][0, :n_valid], outs[1][2][l][0, :n_valid]
        assert torch.equal(ma, mb) and bool(ma.all())
        assert not bool(outs[1][2][l][0, n_valid:].any())
        for part in (0, 1):
            a, b = outs[0][part][l][0, :n_valid], outs[1][part][l][0, :n_valid]
            worst = max(worst, float((a - b).abs().max()))
    assert worst <= 2e-5, worst
