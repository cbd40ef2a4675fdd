"""Pin the CPU oracle (oracle/) against fixtures produced by the real reference
(tests/golden/make_golden.py).  CPU only."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import Golden, ROOT, load_pkg

sys.path.insert(0, ROOT)
from oracle import decafnet_ref as R  # noqa: E402
from oracle import nms_oracle  # noqa: E402

TOL = dict(rtol=1e-5, atol=1e-5)


def close(a, b, **kw):
    kw = {**TOL, **kw}
    assert a.shape == b.shape, (a.shape, b.shape)
    torch.testing.assert_close(a, b, **kw)


# ------------------------------------------------------------------ G1
@pytest.fixture(scope='module')
def ops():
    return Golden('ops.npz')


def test_masked_conv(ops):
    x, mask = ops.t('x'), ops.t('mask')
    w = ops.sub('conv_k3/w/')
    y, m = R.masked_conv1d(x, mask, w['conv.weight'], None, 1, 1)
    close(y, ops.t('conv_k3/y'))
    w = ops.sub('conv_dw_s2/w/')
    y, m = R.masked_conv1d(x, mask, w['conv.weight'], None, 2, 1, groups=x.size(1))
    close(y, ops.t('conv_dw_s2/y'))
    assert torch.equal(m, ops.t('conv_dw_s2/ymask'))
    w = ops.sub('conv_1x1/w/')
    y, _ = R.masked_conv1d(x, mask, w['conv.weight'], w['conv.bias'])
    close(y, ops.t('conv_1x1/y'))


def test_layer_norm(ops):
    x = ops.t('x')
    close(R.channel_layer_norm(x, ops.t('ln/w/weight'), ops.t('ln/w/bias')), ops.t('ln/y'))
    close(R.channel_layer_norm(x), ops.t('ln_noaffine/y'))


def test_max_pool(ops):
    y, m = R.masked_max_pool1d(ops.t('x'), ops.t('mask'))
    close(y, ops.t('maxpool/y'))
    assert torch.equal(m, ops.t('maxpool/ymask'))


def test_mha_global(ops):
    sd = {'a.' + k: v for k, v in ops.sub('mha_global/w/').items()}
    y = R.mha_global(sd, 'a', ops.t('x'), ops.t('kv'), ops.t('kv_mask'), 4)
    close(y, ops.t('mha_global/y'))


@pytest.mark.parametrize('w', [5, 9, 19])
def test_mha_local(ops, w):
    sd = {'a.' + k: v for k, v in ops.sub(f'mha_local{w}/w/').items()}
    x = ops.t('x')
    y = R.mha_local(sd, 'a', x, x, x, ops.t('mask'), 4, w)
    close(y, ops.t(f'mha_local{w}/y'))


@pytest.mark.parametrize('s', [0, 1, 2])
def test_encoder(ops, s):
    sd = {'e.' + k: v for k, v in ops.sub(f'enc_s{s}/w/').items()}
    y, m = R.transformer_encoder(sd, 'e', ops.t('x'), ops.t('mask'), s, 4, 9 if s else 0)
    close(y, ops.t(f'enc_s{s}/y'))
    if s:
        assert torch.equal(m, ops.t(f'enc_s{s}/ymask'))


def test_decoder(ops):
    sd = {'d.' + k: v for k, v in ops.sub('dec/w/').items()}
    y, _ = R.transformer_decoder(sd, 'd', ops.t('x'), ops.t('mask'), ops.t('kv'), ops.t('kv_mask'), 4)
    close(y, ops.t('dec/y'))


def test_ops64_wide_text_stream():
    """the same two blocks with a 64-wide key / value stream (the widths the HIP GEMMs take): ops64.npz"""
    o = Golden('ops64.npz')
    sd = {'a.' + k: v for k, v in o.sub('mha_global/w/').items()}
    close(R.mha_global(sd, 'a', o.t('x'), o.t('kv'), o.t('kv_mask'), 4), o.t('mha_global/y'))
    sd = {'d.' + k: v for k, v in o.sub('dec/w/').items()}
    y, _ = R.transformer_decoder(sd, 'd', o.t('x'), o.t('mask'), o.t('kv'), o.t('kv_mask'), 4)
    close(y, o.t('dec/y'))


def _ops256_weights(pkg, o, block):
    meta = o.js('meta')[block]
    return pkg.synth.make_state_dict({k: tuple(v) for k, v in meta['shapes'].items()}, meta['seed'])


def test_ops256_probe_width_blocks(pkg):
    """TransformerDecoder and TransformerEncoder (stride 1 / 2, window 9) at the probe width E = 256, four 64-channel heads:
    ops256.npz (the reference's outputs; weights regenerated from the recorded shapes and seed)"""
    o = Golden('ops256.npz')
    x, mask = o.t('x'), o.t('mask')
    sd = {'d.' + k: v for k, v in _ops256_weights(pkg, o, 'dec').items()}
    y, _ = R.transformer_decoder(sd, 'd', x, mask, o.t('kv'), o.t('kv_mask'), 4)
    close(y, o.t('dec/y'), rtol=2e-5, atol=2e-5)
    for s in (1, 2):
        sd = {'e.' + k: v for k, v in _ops256_weights(pkg, o, f'enc_s{s}').items()}
        y, m = R.transformer_encoder(sd, 'e', x, mask, s, 4, 9)
        close(y, o.t(f'enc_s{s}/y'), rtol=2e-5, atol=2e-5)
        assert torch.equal(m, o.t(f'enc_s{s}/ymask'))


def test_tcn(ops):
    sd = {'r.' + k: v for k, v in ops.sub('tcn/w/').items()}
    y = R.tcn_refine(sd, 'r', ops.t('tcn/x'), ops.t('mask'), 4)
    close(y, ops.t('tcn/y'))


# ------------------------------------------------------------------ G2
def test_gate_cases():
    g = Golden('gate.npz')
    for i, c in enumerate(g.js('cases')):
        correl, want = g.t(f'c{i}/correl'), g.t(f'c{i}/gate')
        got = R.topk_block_gate(correl, c['vid_len'], c['sn'], c['sratio'])
        full = torch.zeros(c['T'])
        full[:c['vid_len']] = got
        assert torch.equal(full.to(torch.uint8), want), c
        # closed form used by the HIP kernel: sequential block means, stable rank, f32 nearest index
        L, sn = c['vid_len'], c['sn']
        n = (L + sn - 1) // sn
        pooled = torch.empty(n)
        for b in range(n):
            seg = correl[b * sn:min((b + 1) * sn, L)]
            acc = torch.zeros((), dtype=torch.float32)
            for v in seg:
                acc = acc + v
            pooled[b] = acc / float(len(seg))
        got2 = R.gate_reference_formula(pooled, L, c['sratio']) if n < 1200 and L < 20000 else None
        if got2 is not None:
            full2 = torch.zeros(c['T'])
            full2[:L] = got2
            assert torch.equal(full2.to(torch.uint8), want), ('closed form', c)


# ------------------------------------------------------------------ G3
@pytest.mark.parametrize('name', ['c1', 'pe', 'nomsf', 'scat', 'sfonly', 'affine', 'stride2', 'stride4', 'pool', 'pool_stride2'])
def test_end_to_end(name):
    g = Golden(f'e2e_{name}.npz')
    pkg = load_pkg()
    meta, kw = g.js('meta'), g.js('opt_kwargs')
    opt = pkg.config.make_opt(**kw)
    shapes = g.js('shapes')
    sd = pkg.synth.make_state_dict(shapes, meta['wseed'])
    chk = torch.stack([sum(v.double().sum() for v in sd.values()), sum(v.double().abs().sum() for v in sd.values())])
    torch.testing.assert_close(chk, g.t('weight_checksum'), rtol=1e-12, atol=0)
    inp = pkg.synth.make_inputs(meta.get('feat_dim', kw['D']), meta['T'], meta['vid_len'], meta['nq'], kw['text_in'], meta['lq'], meta['iseed'])
    texts, tmasks = [], []
    for q, tok in enumerate(inp['tokens']):
        t, m = R.encode_text(sd, opt.model, tok[None], torch.ones(1, 1, tok.size(-1), dtype=torch.bool))
        close(t, g.t(f'q{q}/text'))
        assert torch.equal(m, g.t(f'q{q}/text_mask'))
        texts.append(t)
        tmasks.append(m)
    logits, offsets, masks, inter = R.forward_eval(sd, opt.model, inp['vid'], inp['shallow_vid'], inp['vid_masks'],
                                                   texts, inp['text_cls'], tmasks, return_intermediates=True)
    for q in range(meta['nq']):
        close(inter['per_query'][q]['vid_map'], g.t(f'q{q}/vid_map'), atol=2e-5)
        close(inter['per_query'][q]['fused'], g.t(f'q{q}/fused'), atol=5e-5)
        for l in range(kw['n_levels']):
            close(logits[q][l], g.t(f'q{q}/l{l}/logits'), atol=1e-4, rtol=1e-4)
            close(offsets[q][l], g.t(f'q{q}/l{l}/offsets'), atol=1e-4, rtol=1e-4)
            assert torch.equal(masks[q][l], g.t(f'q{q}/l{l}/mask'))
    pts = R.generate_points(kw['max_seq_len'] * 10, kw['n_levels'], 4, 0.5)
    for l in range(kw['n_levels']):
        want = g.t(f'points/l{l}')
        assert torch.equal(pts[l][:len(want)], want)


# ------------------------------------------------------------------ G4
def test_postproc():
    g = Golden('postproc.npz')
    meta = g.js('meta')
    L, T0 = meta['L'], meta['T0']
    pts = R.generate_points(T0, L, 4, 0.5)
    logits = [g.t(f'l{l}/logits') for l in range(L)]
    offsets = [g.t(f'l{l}/offsets') for l in range(L)]
    masks = [g.t(f'l{l}/mask') for l in range(L)]
    segs, scores = R.collect_segments(pts, logits, offsets, masks)
    close(segs, g.t('segs'), atol=0, rtol=0)
    close(scores, g.t('scores'), atol=0, rtol=0)
    for k, cfg in g.js('nms_cfgs').items():
        s, c = R.batched_nms(segs.clone(), scores.clone(), **cfg)
        close(s, g.t(f'{k}/segs'), atol=1e-4)
        close(c, g.t(f'{k}/scores'), atol=1e-6)
        sec = R.to_seconds(s, 1, 16, 32, 30.0, 1000.0)
        close(sec, g.t(f'{k}/seconds'), atol=1e-4)


def test_postproc_ext_scores():
    """_collect_segments with external per-clip scores (worker_v2.py:1150-1156)"""
    g = Golden('postproc_ext.npz')
    meta = g.js('meta')
    L, T0 = meta['L'], meta['T0']
    pts = R.generate_points(T0, L, 4, 0.5)
    segs, scores = R.collect_segments(pts, [g.t(f'l{l}/logits') for l in range(L)], [g.t(f'l{l}/offsets') for l in range(L)],
                                      [g.t(f'l{l}/mask') for l in range(L)], pre_nms_topk=meta['pre_nms_topk'], ext_scores=g.t('ext'))
    close(segs, g.t('segs'), atol=0, rtol=0)
    close(scores, g.t('scores'), atol=0, rtol=0)


# ------------------------------------------------------------------ G3b: training-mode forward values + point losses
def test_training_forward_and_losses_match_reference():
    """oracle forward_train (model.py:567-632 with dropout 0) and the loss restatements (loss.py) against the reference's own
    train()-mode forward and loss functions (tests/golden/train.npz)"""
    g = Golden('train.npz')
    meta, kw = g.js('meta'), g.js('opt_kwargs')
    pkg = load_pkg()
    sd = pkg.synth.make_state_dict(g.js('shapes'), meta['wseed'])
    opt = pkg.config.make_opt(**kw)
    out4 = R.forward_train(sd, opt.model, g.t('vid'), g.t('shallow'), g.t('vid_masks'), g.t('tokens'), g.t('token_masks'),
                           g.t('text_cls'), meta['sizes'])
    L = kw['n_levels']
    for part, name in zip(out4, ('logits1', 'logits2', 'offsets', 'masks')):
        for l in range(L):
            want = g.t(f'{name}/l{l}')
            if name == 'masks':
                assert torch.equal(part[l], want)
            else:
                torch.testing.assert_close(part[l], want, rtol=1e-4, atol=1e-4)
    l1 = torch.cat([g.t(f'logits1/l{l}') for l in range(L)], 1)
    l2 = torch.cat([g.t(f'logits2/l{l}') for l in range(L)], 1)
    off = torch.cat([g.t(f'offsets/l{l}') for l in range(L)], 1)
    msk = torch.cat([g.t(f'masks/l{l}') for l in range(L)], 1)
    gt_labels, gt_offsets = g.t('gt_labels'), g.t('gt_offsets')
    pos = gt_labels & msk
    assert int(pos.sum()) == int(g.t('loss/n_pos'))
    lab = gt_labels.float() * 0.8 + 0.1
    tol = dict(rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(R.sigmoid_focal_loss(l1[msk], lab[msk], alpha=0.5).sum(), g.t('loss/focal1_sum'), **tol)
    torch.testing.assert_close(R.sigmoid_focal_loss(l2[msk], lab[msk], alpha=0.5).sum(), g.t('loss/focal2_sum'), **tol)
    torch.testing.assert_close(R.sigmoid_focal_loss(l2, lab, alpha=0.5), g.t('loss/focal2_none'), **tol)
    torch.testing.assert_close(R.sigmoid_focal_loss(l2[msk], gt_labels.float()[msk], alpha=-1.0, gamma=1.5, smoothing=False).mean(),
                               g.t('loss/focal2_nosmooth_mean'), **tol)
    torch.testing.assert_close(R.ctr_iou_loss(off[pos], gt_offsets[pos], 'diou').sum(), g.t('loss/diou_sum'), **tol)
    torch.testing.assert_close(R.ctr_iou_loss(off[pos], gt_offsets[pos], 'giou').sum(), g.t('loss/giou_sum'), **tol)
    torch.testing.assert_close(R.ctr_iou_loss(off.reshape(-1, 2), gt_offsets.reshape(-1, 2), 'diou'), g.t('loss/diou_none'), **tol)
    torch.testing.assert_close(R.ctr_iou_loss(off[pos], gt_offsets[pos], 'giou').mean(), g.t('loss/giou_mean'), **tol)


@pytest.mark.parametrize('name', ['late', 'early', 'early_single'])
def test_training_forward_of_the_single_head_classes_matches_reference(name):
    """oracle forward_train_single_head against the reference's train()-mode forward of PtTransformer / PtTransformerEarlyFusion
    (model.py:83-147, :320-362; tests/golden/train_secondary.npz: two videos, 2 + 1 queries, padded tokens)"""
    g = Golden('train_secondary.npz')
    case, meta = g.js('cases')[name], g.js('meta')
    pkg = load_pkg()
    kw = case['opt_kwargs']
    sd = pkg.synth.make_state_dict(g.js(f'{name}/shapes'), case['wseed'])
    opt = pkg.config.make_opt(**kw)
    kind = 'late' if case['cls'] == 'PtTransformer' else case['cls']
    out3 = R.forward_train_single_head(sd, opt.model, kind, g.t(f'{name}/vid'), g.t(f'{name}/shallow'), g.t(f'{name}/vid_masks'),
                                       g.t(f'{name}/tokens'), g.t(f'{name}/token_masks'), g.t(f'{name}/text_cls'), meta['sizes'])
    for part, pn in zip(out3, ('logits', 'offsets', 'masks')):
        for l in range(kw['n_levels']):
            want = g.t(f'{name}/{pn}/l{l}')
            assert part[l].shape == want.shape
            if pn == 'masks':
                assert torch.equal(part[l], want)
            else:
                torch.testing.assert_close(part[l], want, rtol=1e-4, atol=1e-4)


# ------------------------------------------------------------------ G5
def test_nms_known_answers():
    g = Golden('nms_kat.npz')
    for i, c in enumerate(g.js('cases')):
        segs, scores = g.t(f'k{i}/segs'), g.t(f'k{i}/scores')
        idx = nms_oracle.nms(segs, scores, c['iou_thresh'])
        assert torch.equal(idx, g.t(f'k{i}/nms')), c
        for method in (0, 1, 2):
            dets = torch.full((len(segs), 3), -7.0)
            idx = nms_oracle.softnms(segs, scores, dets, c['iou_thresh'], c['sigma'], c['min_score'], method)
            assert torch.equal(idx, g.t(f'k{i}/soft{method}/idx')), (c, method)
            want = g.t(f'k{i}/soft{method}/dets')
            assert torch.equal(dets[:len(idx)], want), (c, method)


def test_nms_known_answers_beyond_4096_candidates():
    """n = 4097 / 6000 / 8192 (more than the HIP kernels keep in LDS): the oracle against the reference's extension"""
    g = Golden('nms_kat_big.npz')
    for i, c in enumerate(g.js('cases')):
        segs, scores = g.t(f'k{i}/segs'), g.t(f'k{i}/scores')
        assert torch.equal(nms_oracle.nms(segs, scores, c['iou_thresh']), g.t(f'k{i}/nms')), c
        for method in (1, 2):
            dets = torch.full((len(segs), 3), -7.0)
            idx = nms_oracle.softnms(segs, scores, dets, c['iou_thresh'], c['sigma'], c['min_score'], method)
            assert torch.equal(idx, g.t(f'k{i}/soft{method}/idx')), (c, method)
            assert torch.equal(dets[:len(idx)], g.t(f'k{i}/soft{method}/dets')), (c, method)


def test_nms_vs_compiled_reference_random():
    """When oracle/_ref holds the reference's own extension, fuzz the C oracle against it."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('oracle_build_ref', os.path.join(ROOT, 'oracle', 'build_ref.py'))
    br = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(br)
    ext = br.load_module()
    if ext is None:
        pytest.skip('oracle/_ref not built')
    g = torch.Generator().manual_seed(99)
    for trial in range(40):
        n = int(torch.randint(1, 300, (1,), generator=g))
        c = torch.rand(n, generator=g) * 100
        ln = torch.rand(n, generator=g) * 30 + 0.1
        segs = torch.stack((c - ln / 2, c + ln / 2), -1).contiguous()
        scores = torch.rand(n, generator=g).contiguous()
        if len(torch.unique(scores)) != n:
            continue
        thr = float(torch.rand(1, generator=g) * 0.8 + 0.05)
        assert torch.equal(nms_oracle.nms(segs, scores, thr), ext.nms(segs, scores, iou_thresh=thr))
        for method in (0, 1, 2):
            d1, d2 = torch.zeros(n, 3), torch.zeros(n, 3)
            i1 = nms_oracle.softnms(segs, scores, d1, thr, 0.5, 0.01, method)
            i2 = ext.softnms(segs, scores, d2, iou_thresh=thr, sigma=0.5, min_score=0.01, method=method)
            assert torch.equal(i1, i2)
            assert torch.equal(d1[:len(i1)], d2[:len(i2)])


@pytest.mark.parametrize('name', ['late', 'second', 'late_scat', 'early', 'early_single'])
def test_secondary_compositions(name):
    """PtTransformer (late fusion, model.py:30-161) and second_fusion=True (model.py:443-444) restatements"""
    g = Golden(f'e2e_{name}.npz')
    pkg = load_pkg()
    meta, kw = g.js('meta'), g.js('opt_kwargs')
    opt = pkg.config.make_opt(**kw)
    sd = pkg.synth.make_state_dict(g.js('shapes'), meta['wseed'])
    inp = pkg.synth.make_inputs(meta.get('feat_dim', kw['D']), meta['T'], meta['vid_len'], meta['nq'], kw['text_in'], meta['lq'], meta['iseed'])
    texts, tmasks = zip(*[R.encode_text(sd, opt.model, t[None], torch.ones(1, 1, t.size(-1), dtype=torch.bool)) for t in inp['tokens']])
    if meta['cls'] == 'PtTransformer':
        out = R.forward_eval_late_fusion(sd, opt.model, inp['vid'], inp['shallow_vid'], inp['vid_masks'], list(texts), inp['text_cls'], list(tmasks))
    elif meta['cls'].startswith('early'):               # PtTransformerEarlyFusion (model.py:163-373)
        out = R.forward_eval_early_fusion(sd, opt.model, inp['vid'], inp['shallow_vid'], inp['vid_masks'], list(texts), inp['text_cls'],
                                          list(tmasks), second_fusion=meta['cls'] == 'early2')
    else:
        out = R.forward_eval(sd, opt.model, inp['vid'], inp['shallow_vid'], inp['vid_masks'], list(texts), inp['text_cls'], list(tmasks),
                             second_fusion=True)
    for q in range(meta['nq']):
        for l in range(kw['n_levels']):
            close(out[0][q][l], g.t(f'q{q}/l{l}/logits'), atol=1e-4, rtol=1e-4)
            close(out[1][q][l], g.t(f'q{q}/l{l}/offsets'), atol=1e-4, rtol=1e-4)
            assert torch.equal(out[2][q][l], g.t(f'q{q}/l{l}/mask'))


def scale_fixture(name):
    """(kw, meta, reference outputs) of tests/golden/e2e_scale_<name>.npz: logits / offsets / masks of every level as the reference
    returned them (masks stored one bit per clip)"""
    g = Golden(f'e2e_scale_{name}.npz')
    meta, kw = g.js('meta'), g.js('opt_kwargs')
    L = kw['n_levels']
    want = ([], [], [])
    for q in range(meta['nq']):
        lg = [g.t(f'q{q}/l{l}/logits') for l in range(L)]
        want[0].append(lg)
        want[1].append([g.t(f'q{q}/l{l}/offsets') for l in range(L)])
        want[2].append([torch.from_numpy(np.unpackbits(g.t(f'q{q}/l{l}/mask').numpy())[:lg[l].numel()].astype(bool)).view(1, 1, -1)
                        for l in range(L)])
    return g, kw, meta, want


@pytest.mark.parametrize('name', ['c3', 'c4'])
def test_oracle_at_bench_scale(name):
    """the oracle against the REFERENCE's outputs where the large-grid kernels run: BASELINE configs[2] (T = 16 384, vid_len 16 001) and
    configs[3] unsharded (T = 65 536, position encoding resampled 8x, 1 084 scoring blocks); model.py:480-565, video_net.py:141-151"""
    g, kw, meta, want = scale_fixture(name)
    pkg = load_pkg()
    opt = pkg.config.make_opt(**kw)
    model_shapes = {k: list(v.shape) for k, v in pkg.modeling.create_model(opt).state_dict().items()}
    sd = pkg.synth.make_state_dict(model_shapes, meta['wseed'])
    chk = torch.stack([sum(v.double().sum() for v in sd.values()), sum(v.double().abs().sum() for v in sd.values())])
    torch.testing.assert_close(chk, g.t('weight_checksum'), rtol=1e-12, atol=0)            # the same weights as the generator's
    inp = pkg.synth.make_inputs(kw['D'], meta['T'], meta['vid_len'], meta['nq'], kw['text_in'], meta['lq'], meta['iseed'])
    torch.testing.assert_close(torch.stack([inp['vid'].double().sum(), inp['shallow_vid'].double().abs().sum()]), g.t('input_checksum'), rtol=1e-12, atol=0)
    texts, tmasks = zip(*[R.encode_text(sd, opt.model, t[None], torch.ones(1, 1, t.size(-1), dtype=torch.bool)) for t in inp['tokens']])
    for q in range(meta['nq']):
        close(texts[q], g.t(f'q{q}/text'), atol=1e-5, rtol=1e-5)
    with torch.no_grad():
        out = R.forward_eval(sd, opt.model, inp['vid'], inp['shallow_vid'], inp['vid_masks'], list(texts), inp['text_cls'], list(tmasks))
    worst = 0.0
    for q in range(meta['nq']):
        for l in range(kw['n_levels']):
            assert torch.equal(out[2][q][l].view(-1), want[2][q][l].view(-1))
            worst = max(worst, float((out[0][q][l] - want[0][q][l]).abs().max()), float((out[1][q][l] - want[1][q][l]).abs().max()))
            close(out[0][q][l], want[0][q][l], atol=2e-5, rtol=2e-5)
            close(out[1][q][l], want[1][q][l], atol=2e-5, rtol=2e-5)
    print(f'oracle vs reference at {name}: max |delta| = {worst:.2e}')


def test_text_identity():
    """TextIdentity + AttNPool1D restatement (text_net.py:22-89, blocks.py:396-411) vs the reference, alone and inside a model"""
    g = Golden('text_identity.npz')
    pkg = load_pkg()
    for i, c in enumerate(g.js('cases')):
        shapes = g.js(f't{i}/shapes')
        sd = {'text_net.' + k: v for k, v in pkg.synth.make_state_dict(shapes, 700 + i).items()} if shapes else {}
        cfg = dict(name='identity', max_seq_len=c['max_seq_len'], n_heads=c['n_heads'], use_abs_pe=c['use_abs_pe'], use_bkgd_token=c['use_bkgd_token'])
        y, m = R.text_identity(sd, cfg, g.t(f't{i}/tokens'), g.t(f't{i}/mask'))
        close(y, g.t(f't{i}/out'), atol=1e-5)
        assert torch.equal(m, g.t(f't{i}/out_mask'))
    kw, meta = g.js('model/opt_kwargs'), g.js('model/meta')
    opt = pkg.config.make_opt(**kw)
    sd = pkg.synth.make_state_dict(g.js('model/shapes'), meta['wseed'])
    inp = pkg.synth.make_inputs(kw['D'], meta['T'], meta['vid_len'], meta['nq'], kw['text_in'], meta['lq'], meta['iseed'])
    texts, tmasks = zip(*[R.encode_text(sd, opt.model, t[None], torch.ones(1, 1, t.size(-1), dtype=torch.bool)) for t in inp['tokens']])
    lg, of, mk = R.forward_eval(sd, opt.model, inp['vid'], inp['shallow_vid'], inp['vid_masks'], list(texts), inp['text_cls'], list(tmasks))
    for q in range(meta['nq']):
        close(texts[q], g.t(f'model/q{q}/text'), atol=1e-5)
        for l in range(kw['n_levels']):
            close(lg[q][l], g.t(f'model/q{q}/l{l}/logits'), atol=1e-4, rtol=1e-4)
            close(of[q][l], g.t(f'model/q{q}/l{l}/offsets'), atol=1e-4, rtol=1e-4)
            assert torch.equal(mk[q][l], g.t(f'model/q{q}/l{l}/mask'))


def test_c_oracle_is_clean_under_the_sanitizers():
    """oracle/nms_ref.c under AddressSanitizer + UBSan (`make -C oracle sanitize`: san_fuzz.c runs NMS and the three soft-NMS methods over
    random candidate sets of every small size, heavy overlaps, equal scores and scores below the pruning threshold, on exact-size buffers).
    The HIP kernels' indices are compared with this code bit for bit, so an out-of-bounds read in it would be a wrong checker; GPU
    sanitizers are not available on this pool."""
    import shutil
    import subprocess
    if shutil.which('gcc') is None and shutil.which('cc') is None:
        pytest.skip('no C compiler')
    r = subprocess.run(['make', '-C', os.path.join(ROOT, 'oracle'), 'sanitize'], capture_output=True, text=True, timeout=300)
    if r.returncode != 0 and ('cannot find -lasan' in r.stderr or 'libasan' in r.stderr or 'unrecognized' in r.stderr):
        pytest.skip('the sanitizer runtimes are not installed here')
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'san_fuzz ok' in r.stdout
