"""Generate the golden fixtures under tests/golden/ by running the REAL reference.

Run in the build container only (``/root/reference`` is not present on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference is imported read-only from /root/reference; nothing of it is copied here --
the fixtures hold inputs, seeds, configs and the reference's numerical outputs only.
Modules the reference imports at module level but never touches on the grounding path
(yacs, torchtext, decord, torchvision, wandb, cv2) are absent from this image and are
replaced by inert stubs.

Fixture families:  G1 ops.npz / ops64.npz, G2 gate.npz, G3 e2e_*.npz, G4 postproc.npz,
G5 nms_kat.npz, G6 data_io.npz (feature files, annotations, text-CLS table through the reference's loaders),
G3 at bench scale: e2e_scale_c3.npz / e2e_scale_c4.npz (T = 16 384 / 65 536, outputs only; `make_golden.py scale`).
"""
import importlib
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

pkg = importlib.import_module('cvpr2025-decafnet_amd')
make_opt = pkg.config.make_opt
synth = pkg.synth


class _Stub(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith('__'):
            raise AttributeError(k)
        m = _Stub(self.__name__ + '.' + k)
        sys.modules[m.__name__] = m
        setattr(self, k, m)
        return m

    def __call__(self, *a, **k):
        return None


def install_stubs():
    for name in ['yacs', 'yacs.config', 'torchtext', 'torchtext.data', 'decord', 'torchvision',
                 'torchvision.transforms', 'torchvision.transforms.v2', 'wandb', 'cv2',
                 'torchvision.transforms._transforms_video']:
        if name not in sys.modules:
            sys.modules[name] = _Stub(name)
    sys.modules['yacs.config'].CfgNode = dict
    # the compiled reference extension (oracle/_ref) under its real module name
    spec = importlib.util.spec_from_file_location('oracle_build_ref', os.path.join(ROOT, 'oracle', 'build_ref.py'))
    br = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(br)
    br.build()
    ext = br.load_module()
    assert ext is not None, 'reference NMS extension failed to build'
    sys.modules['nms_1d_cpu_vg'] = ext
    return ext


def npify(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.detach().cpu().numpy()
        elif isinstance(v, (dict, list, tuple)) and not isinstance(v, np.ndarray):
            out[k] = np.frombuffer(json.dumps(v).encode(), dtype=np.uint8)
        else:
            out[k] = np.asarray(v)
    return out


def save(name, d):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **npify(d))
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(d)} arrays')


def rand_module_(m, seed):
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = synth.make_state_dict(shapes, seed)
    m.load_state_dict(sd)
    return sd


# ------------------------------------------------------------------ G1: ops
@torch.no_grad()
def gen_ops():
    from libs.modeling import blocks as B
    from libs.modeling.tcn import TCN
    g = torch.Generator().manual_seed(11)
    out = {}
    E, T = 32, 72
    x = torch.randn(2, E, T, generator=g)
    mask = torch.ones(2, 1, T, dtype=torch.bool)
    mask[0, :, 60:] = False
    mask[1, :, 17:] = False
    out['x'] = x
    out['mask'] = mask

    # MaskedConv1D dense k3 s1 / depthwise k3 s2 / 1x1 with bias
    for tag, kw in [('conv_k3', dict(kernel_size=3, stride=1, padding=1, bias=False)),
                    ('conv_dw_s2', dict(kernel_size=3, stride=2, padding=1, groups=E, bias=False)),
                    ('conv_1x1', dict(kernel_size=1))]:
        m = B.MaskedConv1D(E, E, **kw).eval()
        sd = rand_module_(m, 100 + len(tag))
        y, ym = m(x, mask)
        for k, v in sd.items():
            out[f'{tag}/w/{k}'] = v
        out[f'{tag}/y'] = y
        out[f'{tag}/ymask'] = ym

    m = B.LayerNorm(E).eval()
    sd = rand_module_(m, 7)
    out['ln/w/weight'], out['ln/w/bias'] = sd['weight'], sd['bias']
    out['ln/y'] = m(x)
    out['ln_noaffine/y'] = B.LayerNorm(E, affine=False)(x)

    y, ym = B.masked_max_pool1d(x, mask)
    out['maxpool/y'], out['maxpool/ymask'] = y, ym

    # global cross attention: 33 keys, partially masked
    kv = torch.randn(2, 48, 33, generator=g)
    kv_mask = torch.ones(2, 1, 33, dtype=torch.bool)
    kv_mask[1, :, 20:] = False
    out['kv'], out['kv_mask'] = kv, kv_mask
    m = B.MaskedMHA(E, kv_dim=48, out_dim=2 * E, n_heads=4).eval()
    sd = rand_module_(m, 21)
    for k, v in sd.items():
        out[f'mha_global/w/{k}'] = v
    out['mha_global/y'] = m(x, kv, None, kv_mask)

    for w in (5, 9, 19):
        m = B.MaskedMHA(E, n_heads=4, window_size=w).eval()
        sd = rand_module_(m, 30 + w)
        for k, v in sd.items():
            out[f'mha_local{w}/w/{k}'] = v
        out[f'mha_local{w}/y'] = m(x, x, x, mask)

    for s in (1, 2):
        m = B.TransformerEncoder(E, stride=s, n_heads=4, window_size=9).eval()
        sd = rand_module_(m, 40 + s)
        for k, v in sd.items():
            out[f'enc_s{s}/w/{k}'] = v
        y, ym = m(x, mask)
        out[f'enc_s{s}/y'], out[f'enc_s{s}/ymask'] = y, ym

    m = B.TransformerEncoder(E, stride=0, n_heads=4, window_size=0).eval()     # text-style
    sd = rand_module_(m, 43)
    for k, v in sd.items():
        out[f'enc_s0/w/{k}'] = v
    y, _ = m(x, mask)
    out['enc_s0/y'] = y

    m = B.TransformerDecoder(E, 48, n_heads=4).eval()
    sd = rand_module_(m, 50)
    for k, v in sd.items():
        out[f'dec/w/{k}'] = v
    y, ym = m(x, mask, kv, kv_mask)
    out['dec/y'] = y

    m = TCN(4, 32, 32, num_layers=4, in_map=True).eval()
    sd = rand_module_(m, 60)
    for k, v in sd.items():
        out[f'tcn/w/{k}'] = v
    xin = torch.randn(2, 4, T, generator=g)
    out['tcn/x'] = xin
    out['tcn/y'] = m(xin, mask)
    save('ops.npz', out)


@torch.no_grad()
def gen_ops64():
    """The blocks of ops.npz whose key / value width (48) the MI355X GEMMs cannot take (K % 32 == 0): cross-attention MHA and
    the TransformerDecoder with a 64-wide text stream, on 64-channel clips, so that the HIP path runs them too."""
    from libs.modeling import blocks as B
    g = torch.Generator().manual_seed(12)
    out = {}
    E, T, TEk = 64, 72, 64
    x = torch.randn(2, E, T, generator=g)
    mask = torch.ones(2, 1, T, dtype=torch.bool)
    mask[0, :, 60:] = False
    mask[1, :, 17:] = False
    kv = torch.randn(2, TEk, 33, generator=g)
    kv_mask = torch.ones(2, 1, 33, dtype=torch.bool)
    kv_mask[1, :, 20:] = False
    out['x'], out['mask'], out['kv'], out['kv_mask'] = x, mask, kv, kv_mask
    m = B.MaskedMHA(E, kv_dim=TEk, out_dim=2 * E, n_heads=4).eval()
    sd = rand_module_(m, 121)
    for k, v in sd.items():
        out[f'mha_global/w/{k}'] = v
    out['mha_global/y'] = m(x, kv, None, kv_mask)
    m = B.TransformerDecoder(E, TEk, n_heads=4).eval()
    sd = rand_module_(m, 150)
    for k, v in sd.items():
        out[f'dec/w/{k}'] = v
    y, ym = m(x, mask, kv, kv_mask)
    out['dec/y'] = y
    save('ops64.npz', out)


@torch.no_grad()
def gen_ops256():
    """The reference's TransformerDecoder and TransformerEncoder (stride 1 / 2, window 9) at the PROBE width E = 256 with four
    64-channel heads -- the width the one-kernel attention halves (csrc/dec_chain.hip, csrc/enc_chain.hip) are built for, which the
    32- / 64-channel blocks of ops.npz / ops64.npz cannot reach.  Weights are synth.make_state_dict(shapes, seed): the fixture keeps
    (shapes, seed) and the reference's outputs, the tests regenerate the tensors."""
    from libs.modeling import blocks as B
    g = torch.Generator().manual_seed(13)
    out = {}
    E, T, TEk, Lk = 256, 192, 256, 33
    x = torch.randn(2, E, T, generator=g)
    mask = torch.ones(2, 1, T, dtype=torch.bool)
    mask[0, :, 170:] = False
    mask[1, :, 127:129] = False            # a hole across a 128-row window boundary
    mask[1, :, 150:] = False
    kv = torch.randn(2, TEk, Lk, generator=g)
    kv_mask = torch.ones(2, 1, Lk, dtype=torch.bool)
    kv_mask[1, :, 20:] = False
    out['x'], out['mask'], out['kv'], out['kv_mask'] = x, mask, kv, kv_mask
    meta = {}
    m = B.TransformerDecoder(E, TEk, n_heads=4).eval()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(synth.make_state_dict(shapes, 250))
    y, _ = m(x, mask, kv, kv_mask)
    out['dec/y'] = y
    meta['dec'] = dict(shapes={k: list(v) for k, v in shapes.items()}, seed=250)
    for s_ in (1, 2):
        m = B.TransformerEncoder(E, stride=s_, n_heads=4, window_size=9).eval()
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        m.load_state_dict(synth.make_state_dict(shapes, 240 + s_))
        y, ym = m(x, mask)
        out[f'enc_s{s_}/y'], out[f'enc_s{s_}/ymask'] = y, ym
        meta[f'enc_s{s_}'] = dict(shapes={k: list(v) for k, v in shapes.items()}, seed=240 + s_)
    out['meta'] = meta
    save('ops256.npz', out)


# ------------------------------------------------------------------ G2: gate
@torch.no_grad()
def gen_gate():
    """Drive the reference's in-line gate (model.py:531-541) through a tiny model and read the
    0/1 gate off ``vid_map``'s input: with D=1, norm=False and text_cls=[[1]] the sidekick
    score equals the shallow feature itself, and with vid == 1 channel 0 of vid_map's input
    IS the gate."""
    from libs.modeling.model import PtTransformerEarlyFusionIterative
    out = {}
    cases = []
    g = torch.Generator().manual_seed(5)
    specs = [  # (T_pad, vid_len, sn, sratio)
        (256, 240, 60, 0.3), (256, 250, 60, 0.3), (256, 256, 16, 0.5), (64, 40, 60, 0.3),
        (256, 200, 60, 0.0), (1000, 1000, 60, 0.3), (1024, 1000, 7, 0.3), (16384, 16384, 60, 0.3),
        (16384, 16000, 60, 0.3), (65536, 65530, 60, 0.3), (65536, 65536, 60, 0.3), (256, 120, 60, 0.3),
        (256, 240, 60, 1.0), (128, 128, 64, 0.5),
    ]
    for ci, (T, L, sn, sr) in enumerate(specs):
        opt = make_opt(D=1, E=8, TE=8, text_in=4, n_levels=1, win=3, n_heads=1, sn=sn, sratio=sr,
                       msf=True, norm=False, max_seq_len=T, text_layers=1, fusion_layers=1,
                       n_embd_convs=0, use_abs_pe=False)
        model = PtTransformerEarlyFusionIterative(opt.clone(), second_fusion=False).eval()
        seen = {}
        model.vid_map.register_forward_pre_hook(lambda mod, args: seen.__setitem__('x', args[0].clone()))
        correl = torch.randn(T, generator=g)
        shallow = correl.view(1, 1, T).clone()
        vid = torch.ones(1, 1, T)
        mask = (torch.arange(T) < L).view(1, T)
        text, tmask = model.encode_text(torch.randn(1, 4, 3, generator=g), torch.ones(1, 1, 3, dtype=torch.bool))
        model(vid, shallow, mask, (text,), torch.ones(1, 1), (tmask,), eval=True)
        gate = seen['x'][0, 0]
        out[f'c{ci}/correl'] = correl
        out[f'c{ci}/gate'] = gate.to(torch.uint8)
        cases.append(dict(T=T, vid_len=L, sn=sn, sratio=sr))
    out['cases'] = cases
    save('gate.npz', out)


# ------------------------------------------------------------------ G3: end to end
E2E_CASES = {
    # BASELINE config 1: T=256 (240 valid), D=256, Lq=16, NQ=3, E=128, L=6, w=5
    'c1': dict(opt=dict(D=256, E=128, TE=128, text_in=64, n_levels=6, win=5, n_heads=4, sn=60, sratio=0.3,
                        msf=True, norm=True, max_seq_len=256, text_layers=2, text_max_len=24),
               T=256, vid_len=240, nq=3, lq=16, wseed=2025, iseed=2026),
    # PE resample path (T > max_seq_len), vid_len not a multiple of sn, sn=16
    'pe': dict(opt=dict(D=64, E=64, TE=32, text_in=32, n_levels=4, win=9, n_heads=4, sn=16, sratio=0.3,
                        msf=True, norm=True, max_seq_len=128, text_layers=1, text_max_len=24),
               T=256, vid_len=250, nq=2, lq=8, wseed=31, iseed=32),
    # no multi-scale-fusion: the gate is ANDed into the mask (model.py:544-545); raw-dot scores
    'nomsf': dict(opt=dict(D=64, E=64, TE=32, text_in=32, n_levels=3, win=5, n_heads=2, sn=8, sratio=0.5,
                           msf=False, norm=False, max_seq_len=128, text_layers=1, text_max_len=24),
                  T=128, vid_len=100, nq=2, lq=5, wseed=41, iseed=42),
    # opt.model.scat: the raw sidekick score row is one more vid_map input channel (model.py:413-414,550-551)
    'scat': dict(opt=dict(D=64, E=64, TE=32, text_in=32, n_levels=3, win=5, n_heads=2, sn=8, sratio=0.4,
                          msf=True, scat=True, norm=True, max_seq_len=128, text_layers=1, text_max_len=24),
                 T=128, vid_len=115, nq=2, lq=6, wseed=71, iseed=72),
    # opt.model.sfonly (+ scat): vid_map sees the sidekick features alone (model.py:546-547); vid_net.in_dim * 2 must then
    # equal the feature width, so the features are 64 wide with in_dim = 32
    'sfonly': dict(opt=dict(D=32, E=64, TE=32, text_in=32, n_levels=3, win=5, n_heads=2, sn=8, sratio=0.4,
                            msf=True, scat=True, sfonly=True, norm=True, max_seq_len=128, text_layers=1, text_max_len=24),
                   feat_dim=64, T=128, vid_len=120, nq=2, lq=4, wseed=81, iseed=82),
    # opt.model.fusion.xattn_mode = 'affine' (blocks.py:613-626): the cross-attention output modulates q itself, not LayerNorm(q)
    'affine': dict(opt=dict(D=64, E=64, TE=32, text_in=32, n_levels=4, win=5, n_heads=4, sn=16, sratio=0.3, msf=True, norm=True,
                            max_seq_len=128, text_layers=1, text_max_len=24, xattn_mode='affine'),
                   T=256, vid_len=251, nq=2, lq=6, wseed=33, iseed=34),
    # opt.model.vid_net.stride = 2 (video_net.py:59-74): the first embedding convolution is k5 / stride 2 / padding 2, the pyramid
    # starts at T / 2 (worker_v2.py:285-286: input_vid_len = max_vid_len * vid_stride); position encoding on the halved sequence
    'stride2': dict(opt=dict(D=64, E=64, TE=32, text_in=32, n_levels=3, win=5, n_heads=4, sn=16, sratio=0.3, msf=True, norm=True,
                             max_seq_len=128, text_layers=1, text_max_len=24, vid_stride=2),
                    T=256, vid_len=243, nq=2, lq=6, wseed=101, iseed=102),
    # stride 4: both embedding convolutions downsample; the resampled position encoding (T / 4 > max_seq_len)
    'stride4': dict(opt=dict(D=64, E=64, TE=32, text_in=32, n_levels=3, win=5, n_heads=2, sn=16, sratio=0.4, msf=True, norm=True,
                             max_seq_len=64, text_layers=1, text_max_len=24, vid_stride=4),
                    T=512, vid_len=489, nq=2, lq=5, wseed=103, iseed=104),
    # opt.model.vid_net.pool_only (video_net.py:98-111): every branch layer is one depthwise k3 MaskedConv1D (stride 1, then 2)
    'pool': dict(opt=dict(D=64, E=64, TE=32, text_in=32, n_levels=4, win=5, n_heads=4, sn=16, sratio=0.3, msf=True, norm=True,
                          max_seq_len=128, text_layers=1, text_max_len=24, pool_only=True),
                 T=256, vid_len=250, nq=2, lq=7, wseed=105, iseed=106),
    'pool_stride2': dict(opt=dict(D=64, E=64, TE=32, text_in=32, n_levels=3, win=5, n_heads=4, sn=8, sratio=0.5, msf=False, norm=True,
                                  max_seq_len=128, text_layers=1, text_max_len=24, pool_only=True, vid_stride=2, n_stem=1),
                         T=256, vid_len=231, nq=2, lq=4, wseed=107, iseed=108),
}
ONLY = set(filter(None, os.environ.get('ONLY', '').split(',')))


# secondary compositions of the same blocks: late fusion (PtTransformer) and second_fusion=True
E2E_VARIANTS = {
    'late': dict(cls='PtTransformer', opt=dict(D=64, E=64, TE=32, text_in=32, n_levels=4, win=5, n_heads=4, sn=16, sratio=0.3,
                                               msf=True, norm=True, max_seq_len=128, text_layers=1, text_max_len=24),
                 T=256, vid_len=230, nq=2, lq=7, wseed=51, iseed=52),
    'second': dict(cls='iter2', opt=dict(D=64, E=64, TE=32, text_in=32, n_levels=4, win=5, n_heads=4, sn=16, sratio=0.3,
                                         msf=True, norm=True, max_seq_len=256, text_layers=1, text_max_len=24),
                   T=256, vid_len=256, nq=2, lq=5, wseed=61, iseed=62),
    # PtTransformer without msf and with scat (model.py:43-48,124-129): embd_fc is (E, D+1, 1), the gate goes into the mask
    'late_scat': dict(cls='PtTransformer', opt=dict(D=64, E=64, TE=32, text_in=32, n_levels=3, win=5, n_heads=4, sn=8, sratio=0.5,
                                                    msf=False, scat=True, norm=True, max_seq_len=128, text_layers=1, text_max_len=24),
                      T=128, vid_len=101, nq=2, lq=5, wseed=91, iseed=92),
    # PtTransformerEarlyFusion (model.py:163-373): early fusion without the refinement stage; second_fusion defaults to True
    'early': dict(cls='early2', opt=dict(D=64, E=64, TE=32, text_in=32, n_levels=4, win=5, n_heads=4, sn=16, sratio=0.3,
                                         msf=True, norm=True, max_seq_len=128, text_layers=1, text_max_len=24),
                  T=256, vid_len=243, nq=2, lq=6, wseed=53, iseed=54),
    'early_single': dict(cls='early1', opt=dict(D=64, E=64, TE=32, text_in=32, n_levels=3, win=5, n_heads=4, sn=8, sratio=0.5,
                                                msf=False, scat=True, norm=False, max_seq_len=128, text_layers=1, text_max_len=24),
                         T=128, vid_len=120, nq=2, lq=5, wseed=55, iseed=56),
}


@torch.no_grad()
def gen_e2e_variants():
    from libs.modeling.model import PtTransformerEarlyFusionIterative, PtTransformer, PtTransformerEarlyFusion
    for name, c in E2E_VARIANTS.items():
        if ONLY and name not in ONLY:
            continue
        opt = make_opt(**c['opt'])
        if c['cls'] == 'PtTransformer':
            model = PtTransformer(opt.clone()).eval()
        elif c['cls'] in ('early1', 'early2'):
            model = PtTransformerEarlyFusion(opt.clone(), second_fusion=c['cls'] == 'early2').eval()
        else:
            model = PtTransformerEarlyFusionIterative(opt.clone(), second_fusion=True).eval()
        shapes = {k: list(v.shape) for k, v in model.state_dict().items()}
        sd = synth.make_state_dict(shapes, c['wseed'])
        model.load_state_dict(sd)
        inp = synth.make_inputs(c['opt']['D'], c['T'], c['vid_len'], c['nq'], c['opt']['text_in'], c['lq'], c['iseed'])
        texts, tmasks = [], []
        for tok in inp['tokens']:
            t, m = model.encode_text(tok[None], torch.ones(1, 1, tok.size(-1), dtype=torch.bool))
            texts.append(t)
            tmasks.append(m)
        logits, offsets, masks = model(inp['vid'], inp['shallow_vid'], inp['vid_masks'], tuple(texts), inp['text_cls'], tuple(tmasks), eval=True)
        out = dict(opt_kwargs=c['opt'], meta=dict(T=c['T'], vid_len=c['vid_len'], nq=c['nq'], lq=c['lq'], wseed=c['wseed'],
                                                  iseed=c['iseed'], cls=c['cls']), shapes=shapes)
        for q in range(c['nq']):
            for l in range(len(logits[q])):
                out[f'q{q}/l{l}/logits'] = logits[q][l]
                out[f'q{q}/l{l}/offsets'] = offsets[q][l]
                out[f'q{q}/l{l}/mask'] = masks[q][l]
        save(f'e2e_{name}.npz', out)


@torch.no_grad()
def gen_e2e():
    from libs.modeling.model import PtTransformerEarlyFusionIterative, PtGenerator
    for name, c in E2E_CASES.items():
        if ONLY and name not in ONLY:
            continue
        opt = make_opt(**c['opt'])
        model = PtTransformerEarlyFusionIterative(opt.clone(), second_fusion=False).eval()
        shapes = {k: list(v.shape) for k, v in model.state_dict().items()}
        sd = synth.make_state_dict(shapes, c['wseed'])
        model.load_state_dict(sd)
        inp = synth.make_inputs(c.get('feat_dim', c['opt']['D']), c['T'], c['vid_len'], c['nq'], c['opt']['text_in'], c['lq'], c['iseed'])
        texts, tmasks = [], []
        for tok in inp['tokens']:
            t, m = model.encode_text(tok[None], torch.ones(1, 1, tok.size(-1), dtype=torch.bool))
            texts.append(t)
            tmasks.append(m)
        seen = {}
        model.vid_map.register_forward_hook(lambda mod, a, o: seen.setdefault('vid_map', []).append(o[0].clone()))
        model.fusion.register_forward_hook(lambda mod, a, o: seen.setdefault('fused', []).append(o[0].clone()))
        logits, offsets, masks = model(inp['vid'], inp['shallow_vid'], inp['vid_masks'], tuple(texts),
                                       inp['text_cls'], tuple(tmasks), eval=True)
        out = dict(opt_kwargs=c['opt'], meta=dict(T=c['T'], vid_len=c['vid_len'], nq=c['nq'], lq=c['lq'],
                                                  wseed=c['wseed'], iseed=c['iseed'], feat_dim=c.get('feat_dim', c['opt']['D'])),
                   shapes=shapes,
                   weight_checksum=torch.stack([sum(v.double().sum() for v in sd.values()),
                                                sum(v.double().abs().sum() for v in sd.values())]))
        for q in range(c['nq']):
            out[f'q{q}/text'] = texts[q]
            out[f'q{q}/text_mask'] = tmasks[q]
            out[f'q{q}/vid_map'] = seen['vid_map'][q]
            out[f'q{q}/fused'] = seen['fused'][q]
            for l in range(len(logits[q])):
                out[f'q{q}/l{l}/logits'] = logits[q][l]
                out[f'q{q}/l{l}/offsets'] = offsets[q][l]
                out[f'q{q}/l{l}/mask'] = masks[q][l]
        # point generator (PtGenerator, model.py:668-743) sized 10x as the evaluator does
        pg = PtGenerator(max_seq_len=c['opt']['max_seq_len'] * 10, num_fpn_levels=c['opt']['n_levels'],
                         regression_range=4, sigma=0.5)
        pts = pg([m.size(-1) for m in masks[0]])
        for l, p in enumerate(pts):
            out[f'points/l{l}'] = p
        save(f'e2e_{name}.npz', out)


# Outputs-only fixtures at the scale the large-grid kernels run at: weights and inputs regenerate from the seeds
# (synth.make_state_dict / make_inputs), the file holds the reference's logits / offsets / masks of every level and its encoded text.
E2E_SCALE = {
    # BASELINE configs[2]: the probe configuration of tests/test_gpu_e2e.py::test_probe_config_vs_oracle[16384]
    'c3': dict(opt=dict(D=1024, E=256, TE=256, text_in=128, n_levels=8, win=9, n_heads=4, sn=60, sratio=0.3, msf=True, norm=True,
                        max_seq_len=2048, text_layers=2, text_max_len=48), T=16384, vid_len=16001, nq=1, lq=32, wseed=7, iseed=8),
    # BASELINE configs[3] unsharded: test_config4_unsharded_T65536_vs_oracle (position encoding resampled 8x, video_net.py:147-150)
    'c4': dict(opt=dict(D=1024, E=256, TE=256, text_in=128, n_levels=8, win=9, n_heads=4, sn=60, sratio=0.3, msf=True, norm=True,
                        max_seq_len=8192, text_layers=2, text_max_len=48), T=65536, vid_len=65000, nq=1, lq=32, wseed=11, iseed=12),
}


@torch.no_grad()
def gen_e2e_scale():
    from libs.modeling.model import PtTransformerEarlyFusionIterative
    for name, c in E2E_SCALE.items():
        if ONLY and name not in ONLY:
            continue
        opt = make_opt(**c['opt'])
        model = PtTransformerEarlyFusionIterative(opt.clone(), second_fusion=False).eval()
        shapes = {k: list(v.shape) for k, v in model.state_dict().items()}
        sd = synth.make_state_dict(shapes, c['wseed'])
        model.load_state_dict(sd)
        inp = synth.make_inputs(c['opt']['D'], c['T'], c['vid_len'], c['nq'], c['opt']['text_in'], c['lq'], c['iseed'])
        texts, tmasks = zip(*[model.encode_text(tok[None], torch.ones(1, 1, tok.size(-1), dtype=torch.bool)) for tok in inp['tokens']])
        logits, offsets, masks = model(inp['vid'], inp['shallow_vid'], inp['vid_masks'], tuple(texts), inp['text_cls'], tuple(tmasks), eval=True)
        out = dict(opt_kwargs=c['opt'], meta=dict(T=c['T'], vid_len=c['vid_len'], nq=c['nq'], lq=c['lq'], wseed=c['wseed'], iseed=c['iseed']),
                   weight_checksum=torch.stack([sum(v.double().sum() for v in sd.values()), sum(v.double().abs().sum() for v in sd.values())]),
                   input_checksum=torch.stack([inp['vid'].double().sum(), inp['shallow_vid'].double().abs().sum()]))
        for q in range(c['nq']):
            out[f'q{q}/text'] = texts[q]
            for l in range(len(logits[q])):
                out[f'q{q}/l{l}/logits'] = logits[q][l]
                out[f'q{q}/l{l}/offsets'] = offsets[q][l]
                out[f'q{q}/l{l}/mask'] = np.packbits(masks[q][l].numpy().astype(np.uint8))      # (a prefix of ones: one bit per clip)
        save(f'e2e_scale_{name}.npz', out)


@torch.no_grad()
def gen_text_identity():
    """TextIdentity (text_net.py:22-89) in its three shapes -- embedding + attention-pooled token, position encoding with
    padded tokens, pure identity -- and one model whose text_net is 'identity' (reference encode_text + forward)."""
    from libs.modeling.text_net import make_text_net
    from libs.modeling.model import PtTransformerEarlyFusionIterative
    out = {}
    cases = [dict(in_dim=48, embd_dim=64, max_seq_len=16, n_heads=4, use_abs_pe=False, use_bkgd_token=True, lq=9, valid=9),
             dict(in_dim=48, embd_dim=64, max_seq_len=8, n_heads=2, use_abs_pe=True, use_bkgd_token=True, lq=12, valid=7),
             dict(in_dim=64, embd_dim=None, max_seq_len=16, n_heads=4, use_abs_pe=False, use_bkgd_token=False, lq=5, valid=5)]
    for i, c in enumerate(cases):
        kw = {k: v for k, v in c.items() if k not in ('lq', 'valid')}
        net = make_text_net(dict(name='identity', **kw)).eval()
        shapes = {k: list(v.shape) for k, v in net.state_dict().items()}
        sd = synth.make_state_dict(shapes, 700 + i) if shapes else {}
        net.load_state_dict(sd)
        g = torch.Generator().manual_seed(710 + i)
        tok = torch.randn(1, c['in_dim'], c['lq'], generator=g)
        mask = (torch.arange(c['lq']) < c['valid']).view(1, 1, -1)
        y, m = net(tok, mask)
        out[f't{i}/tokens'], out[f't{i}/mask'], out[f't{i}/out'], out[f't{i}/out_mask'] = tok, mask, y, m
        out[f't{i}/shapes'] = shapes
    out['cases'] = cases
    # a model with the identity text path
    kw = dict(D=64, E=64, TE=32, text_in=32, n_levels=3, win=5, n_heads=2, sn=8, sratio=0.4, msf=True, norm=True,
              max_seq_len=128, text_max_len=24, text_name='identity')
    opt = make_opt(**kw)
    model = PtTransformerEarlyFusionIterative(opt.clone(), second_fusion=False).eval()
    shapes = {k: list(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(synth.make_state_dict(shapes, 720))
    inp = synth.make_inputs(64, 128, 111, 2, 32, 6, 721)
    texts, tmasks = zip(*[model.encode_text(t[None], torch.ones(1, 1, t.size(-1), dtype=torch.bool)) for t in inp['tokens']])
    logits, offsets, masks = model(inp['vid'], inp['shallow_vid'], inp['vid_masks'], tuple(texts), inp['text_cls'], tuple(tmasks), eval=True)
    out['model/opt_kwargs'] = kw
    out['model/meta'] = dict(T=128, vid_len=111, nq=2, lq=6, wseed=720, iseed=721)
    out['model/shapes'] = shapes
    for q in range(2):
        out[f'model/q{q}/text'], out[f'model/q{q}/text_mask'] = texts[q], tmasks[q]
        for l in range(3):
            out[f'model/q{q}/l{l}/logits'], out[f'model/q{q}/l{l}/offsets'], out[f'model/q{q}/l{l}/mask'] = logits[q][l], offsets[q][l], masks[q][l]
    save('text_identity.npz', out)


# ------------------------------------------------------------------ G4: post-processing
@torch.no_grad()
def gen_postproc():
    """Evaluator._collect_segments -> batched_nms -> seconds (worker_v2.py:1063-1187) on
    synthetic head outputs with many candidates."""
    from libs.worker_v2 import Evaluator
    from libs.nms.nms import batched_nms
    from libs.modeling.model import PtGenerator
    g = torch.Generator().manual_seed(77)
    T0, L = 2048, 6
    ev = Evaluator.__new__(Evaluator)
    ev.pre_nms_topk, ev.pre_nms_thresh, ev.seg_len_thresh = 2000, 0.001, 0.1
    ev.vid_stride = 1
    pts = PtGenerator(max_seq_len=T0, num_fpn_levels=L, regression_range=4, sigma=0.5)([T0 >> l for l in range(L)])
    vid_len = 1900
    logits, offsets, masks = [], [], []
    for l in range(L):
        n = T0 >> l
        logits.append(torch.randn(1, n, generator=g) * 2.0 - 1.0)
        offsets.append(torch.rand(1, n, 2, generator=g) * 6.0)
        valid = (vid_len + (1 << l) - 1) >> l
        masks.append((torch.arange(n) < valid).view(1, n))
    # make some offsets tiny so the seg_len_thresh filter fires
    offsets[0][0, ::7] = 0.01
    segs, scores = ev._collect_segments(pts, logits, offsets, masks, None)
    out = dict(meta=dict(T0=T0, L=L, vid_len=vid_len))
    for l in range(L):
        out[f'l{l}/logits'], out[f'l{l}/offsets'], out[f'l{l}/mask'] = logits[l], offsets[l], masks[l]
    out['segs'], out['scores'] = segs, scores
    cfgs = dict(soft_default=dict(iou_thresh=0.1, min_score=0.001, max_num_segs=5, mode='soft_nms', sigma=0.9, voting_thresh=0.95),
                soft_novote=dict(iou_thresh=0.1, min_score=0.001, max_num_segs=50, mode='soft_nms', sigma=0.5, voting_thresh=0.0),
                hard_vote=dict(iou_thresh=0.5, min_score=0.001, max_num_segs=20, mode='nms', sigma=0.9, voting_thresh=0.75),
                hard_novote=dict(iou_thresh=0.3, min_score=0.0, max_num_segs=100, mode='nms', sigma=0.9, voting_thresh=0.0))
    out['nms_cfgs'] = cfgs
    for k, cfg in cfgs.items():
        s, c = batched_nms(segs.clone(), scores.clone(), **cfg)
        out[f'{k}/segs'], out[f'{k}/scores'] = s, c
        sec = s.clone()
        if len(sec) > 0:
            sec = sec * 1
            sec = (sec * 16 + 0.5 * 32) / 30.0
            sec = torch.clamp(sec, min=0, max=1000.0)
        out[f'{k}/seconds'] = sec
    # metric loop of Evaluator.run (worker_v2.py:857-878, 890-901) on the soft_novote results against synthetic targets
    from libs.train_utils import iou
    import numpy as np
    segs, scores = out['soft_novote/seconds'], out['soft_novote/scores']
    ranks, iou_threshs = (1, 5), np.array((0.3, 0.5))
    targets = [(float(segs[0, 0]) - 1.0, float(segs[0, 1]) + 2.0), (float(segs[3, 0]), float(segs[3, 1])), (500.0, 510.0)]
    counts = np.zeros((len(ranks), len(iou_threshs)))
    iou_rows = []
    for tgt in targets:
        idx = scores.argsort(descending=True)
        s = segs[idx[:max(ranks)]]
        t = torch.as_tensor(tgt, dtype=torch.float).expand(len(s), -1)
        iou_topk = iou(s, t)
        iou_n = np.array([iou_topk[:i].max().item() if len(iou_topk[:i]) > 0 else 0 for i in ranks])
        counts += (iou_n[:, None] >= iou_threshs[None])
        iou_rows.append(iou_topk)
    out['metric/targets'] = torch.tensor(targets)
    out['metric/iou_topk'] = torch.stack(iou_rows)
    out['metric/counts'] = torch.from_numpy(counts)
    save('postproc.npz', out)


@torch.no_grad()
def gen_postproc_ext():
    """Evaluator._collect_segments with ext_scores (worker_v2.py:1150-1156): the external per-clip scores multiply the
    level scores and are max-pooled (k3, s2, p1) down the pyramid."""
    from libs.worker_v2 import Evaluator
    from libs.modeling.model import PtGenerator
    g = torch.Generator().manual_seed(177)
    T0, L = 1024, 5
    ev = Evaluator.__new__(Evaluator)
    ev.pre_nms_topk, ev.pre_nms_thresh, ev.seg_len_thresh = 500, 0.001, 0.1
    ev.vid_stride = 1
    pts = PtGenerator(max_seq_len=T0, num_fpn_levels=L, regression_range=4, sigma=0.5)([T0 >> l for l in range(L)])
    vid_len = 1000
    logits, offsets, masks = [], [], []
    for l in range(L):
        n = T0 >> l
        logits.append(torch.randn(1, n, generator=g) * 2.0 - 1.0)
        offsets.append(torch.rand(1, n, 2, generator=g) * 6.0)
        valid = (vid_len + (1 << l) - 1) >> l
        masks.append((torch.arange(n) < valid).view(1, n))
    ext = torch.rand(T0, generator=g)
    ext[::5] = 0.0
    segs, scores = ev._collect_segments(pts, [x.clone() for x in logits], offsets, masks, ext.clone())
    out = dict(meta=dict(T0=T0, L=L, vid_len=vid_len, pre_nms_topk=500), ext=ext, segs=segs, scores=scores)
    for l in range(L):
        out[f'l{l}/logits'], out[f'l{l}/offsets'], out[f'l{l}/mask'] = logits[l], offsets[l], masks[l]
    save('postproc_ext.npz', out)


# ------------------------------------------------------------------ G3b: training-mode forward values + point losses
@torch.no_grad()
def gen_train():
    """PtTransformerEarlyFusionIterative.forward(eval=False) (model.py:567-632) in train() mode with every dropout probability 0,
    two videos with 2 + 1 queries (padded text batch, text_size), and the reference's loss functions (loss.py) on its outputs with
    synthetic labels -- the way Trainer.forward_backward combines them (worker_v2.py:441-461)."""
    from libs.modeling.model import PtTransformerEarlyFusionIterative
    from libs.modeling.loss import sigmoid_focal_loss, ctr_giou_loss, ctr_diou_loss
    kw = dict(D=64, E=64, TE=32, text_in=32, n_levels=4, win=5, n_heads=4, sn=8, sratio=0.3, msf=True, norm=True,
              max_seq_len=256, text_layers=2, text_max_len=24)
    opt = make_opt(**kw)
    model = PtTransformerEarlyFusionIterative(opt.clone(), second_fusion=False).train()
    # the refinement TCN hard-codes nn.Dropout(0.5) in every layer (tcn.py:5,13; built without a dropout argument, model.py:424-425):
    # the training forward is stochastic whatever opt says.  Forward VALUES are defined with that dropout at p = 0 as well.
    for mod in model.refine.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    shapes = {k: list(v.shape) for k, v in model.state_dict().items()}
    sd = synth.make_state_dict(shapes, 901)
    model.load_state_dict(sd)
    g = torch.Generator().manual_seed(902)
    bs, T, lq, sizes = 2, 256, 7, [2, 1]
    vid = torch.randn(bs, 64, T, generator=g)
    shallow = torch.randn(bs, 64, T, generator=g)
    lens = [256, 201]
    vid_masks = torch.stack([torch.arange(T) < n for n in lens])
    vid, shallow = vid * vid_masks[:, None], shallow * vid_masks[:, None]
    tokens = torch.randn(sum(sizes), 32, lq, generator=g)
    tok_len = [7, 5, 6]
    token_masks = torch.stack([torch.arange(lq) < n for n in tok_len])[:, None]           # (B', 1, Lq)
    tokens = tokens * token_masks
    text_cls = torch.randn(sum(sizes), 64, generator=g)
    # the padded per-video layout of the training collate (model.py:617-622): (bs, max_k, C, Lq) / (bs, max_k, Lq)
    mk = max(sizes)
    text_pad = torch.zeros(bs, mk, 32, lq)
    mask_pad = torch.zeros(bs, mk, lq, dtype=torch.bool)
    q = 0
    for b, k in enumerate(sizes):
        text_pad[b, :k] = tokens[q:q + k]
        mask_pad[b, :k] = token_masks[q:q + k, 0]
        q += k
    out4 = model(vid, shallow, vid_masks, text_pad, text_cls, mask_pad, text_size=torch.tensor(sizes), eval=False)
    out = dict(opt_kwargs=kw, meta=dict(bs=bs, T=T, lq=lq, sizes=sizes, wseed=901), shapes=shapes, vid=vid, shallow=shallow,
               vid_masks=vid_masks, tokens=tokens, token_masks=token_masks, text_cls=text_cls)
    names = ('logits1', 'logits2', 'offsets', 'masks')
    for part, name in zip(out4, names):
        for l, x in enumerate(part):
            out[f'{name}/l{l}'] = x
    # losses as the Trainer stitches them (worker_v2.py:431-461)
    l1, l2 = torch.cat(out4[0], 1), torch.cat(out4[1], 1)
    off, msk = torch.cat(out4[2], 1), torch.cat(out4[3], 1)
    gt_labels = torch.rand(l1.shape, generator=g) < 0.08
    gt_offsets = torch.rand(off.shape, generator=g) * 20
    pos = torch.logical_and(gt_labels, msk)
    out['gt_labels'], out['gt_offsets'] = gt_labels, gt_offsets
    sm, al = 0.2, 0.5
    lab = gt_labels.float() * (1.0 - sm) + sm / 2
    out['loss/focal1_sum'] = sigmoid_focal_loss(l1[msk], lab[msk], alpha=al, reduction='sum')
    out['loss/focal2_sum'] = sigmoid_focal_loss(l2[msk], lab[msk], alpha=al, reduction='sum')
    out['loss/focal2_none'] = sigmoid_focal_loss(l2, lab, alpha=al, reduction='none')
    out['loss/focal2_nosmooth_mean'] = sigmoid_focal_loss(l2[msk], gt_labels.float()[msk], alpha=-1.0, gamma=1.5, smoothing=False, reduction='mean')
    out['loss/diou_sum'] = ctr_diou_loss(off[pos], gt_offsets[pos], reduction='sum')
    out['loss/giou_sum'] = ctr_giou_loss(off[pos], gt_offsets[pos], reduction='sum')
    out['loss/diou_none'] = ctr_diou_loss(off.reshape(-1, 2), gt_offsets.reshape(-1, 2), reduction='none')
    out['loss/giou_mean'] = ctr_giou_loss(off[pos], gt_offsets[pos], reduction='mean')
    out['loss/n_pos'] = pos.sum()
    save('train.npz', out)


@torch.no_grad()
def gen_train_secondary():
    """The training-mode forward (eval=False, train() mode, every dropout probability 0) of the two classes with one classification head:
    PtTransformer (model.py:110-161) and PtTransformerEarlyFusion with and without the second fusion (model.py:300-373); two videos with
    2 + 1 queries, padded text batch + text_size as the training collate delivers them (model.py:617-622).  One fixture, three cases."""
    from libs.modeling.model import PtTransformer, PtTransformerEarlyFusion
    cases = {
        'late': dict(cls='PtTransformer', opt=dict(D=64, E=64, TE=32, text_in=32, n_levels=4, win=5, n_heads=4, sn=16, sratio=0.3, msf=True, norm=True,
                                                   max_seq_len=256, text_layers=2, text_max_len=24), wseed=911, iseed=912),
        'early': dict(cls='early2', opt=dict(D=64, E=64, TE=32, text_in=32, n_levels=4, win=5, n_heads=4, sn=8, sratio=0.3, msf=True, norm=True,
                                             max_seq_len=256, text_layers=2, text_max_len=24), wseed=921, iseed=922),
        'early_single': dict(cls='early1', opt=dict(D=64, E=64, TE=32, text_in=32, n_levels=3, win=5, n_heads=4, sn=8, sratio=0.5, msf=False, scat=True,
                                                    norm=False, max_seq_len=256, text_layers=1, text_max_len=24), wseed=931, iseed=932),
    }
    out = dict(cases={k: dict(cls=c['cls'], opt_kwargs=c['opt'], wseed=c['wseed']) for k, c in cases.items()})
    bs, T, lq, sizes = 2, 256, 7, [2, 1]
    lens, tok_len = [256, 187], [7, 4, 6]
    out['meta'] = dict(bs=bs, T=T, lq=lq, sizes=sizes)
    for name, c in cases.items():
        opt = make_opt(**c['opt'])
        if c['cls'] == 'PtTransformer':
            model = PtTransformer(opt.clone()).train()
        else:
            model = PtTransformerEarlyFusion(opt.clone(), second_fusion=c['cls'] == 'early2').train()
        shapes = {k: list(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict(synth.make_state_dict(shapes, c['wseed']))
        g = torch.Generator().manual_seed(c['iseed'])
        vid, shallow = torch.randn(bs, 64, T, generator=g), torch.randn(bs, 64, T, generator=g)
        vid_masks = torch.stack([torch.arange(T) < n for n in lens])
        vid, shallow = vid * vid_masks[:, None], shallow * vid_masks[:, None]
        tokens = torch.randn(sum(sizes), 32, lq, generator=g)
        token_masks = torch.stack([torch.arange(lq) < n for n in tok_len])[:, None]
        tokens = tokens * token_masks
        text_cls = torch.randn(sum(sizes), 64, generator=g)
        mk = max(sizes)
        text_pad, mask_pad = torch.zeros(bs, mk, 32, lq), torch.zeros(bs, mk, lq, dtype=torch.bool)
        q = 0
        for b, k in enumerate(sizes):
            text_pad[b, :k], mask_pad[b, :k] = tokens[q:q + k], token_masks[q:q + k, 0]
            q += k
        out3 = model(vid, shallow, vid_masks, text_pad, text_cls, mask_pad, text_size=torch.tensor(sizes), eval=False)
        assert len(out3) == 3
        out[f'{name}/shapes'] = shapes
        for key, val in (('vid', vid), ('shallow', shallow), ('vid_masks', vid_masks), ('tokens', tokens), ('token_masks', token_masks), ('text_cls', text_cls)):
            out[f'{name}/{key}'] = val
        for part, pn in zip(out3, ('logits', 'offsets', 'masks')):
            for l, x in enumerate(part):
                out[f'{name}/{pn}/l{l}'] = x
    save('train_secondary.npz', out)


# ------------------------------------------------------------------ G5: NMS known answers
@torch.no_grad()
def gen_nms(ext):
    g = torch.Generator().manual_seed(1234)
    out, cases = {}, []

    def rand_segs(n, span=1000.0, max_len=60.0):
        c = torch.rand(n, generator=g) * span
        ln = torch.rand(n, generator=g) * max_len + 0.2
        segs = torch.stack((c - ln / 2, c + ln / 2), -1).contiguous()
        scores = torch.rand(n, generator=g)
        # tie-free: perturb until all scores are distinct
        while len(torch.unique(scores)) != n:
            scores = torch.rand(n, generator=g)
        return segs, scores.contiguous()

    def add(tag, segs, scores, thr, sigma=0.9, min_score=0.001):
        i = len(cases)
        out[f'k{i}/segs'], out[f'k{i}/scores'] = segs, scores
        out[f'k{i}/nms'] = ext.nms(segs, scores, iou_thresh=float(thr))
        for method in (0, 1, 2):
            dets = torch.full((len(segs), 3), -7.0)
            idx = ext.softnms(segs, scores, dets, iou_thresh=float(thr), sigma=float(sigma),
                              min_score=float(min_score), method=method)
            out[f'k{i}/soft{method}/idx'] = idx
            out[f'k{i}/soft{method}/dets'] = dets[:len(idx)]
        cases.append(dict(tag=tag, n=int(len(segs)), iou_thresh=thr, sigma=sigma, min_score=min_score))

    add('empty', torch.zeros(0, 2), torch.zeros(0), 0.5)
    add('single', torch.tensor([[1.0, 5.0]]), torch.tensor([0.7]), 0.5)
    add('pair_overlap', torch.tensor([[0.0, 10.0], [1.0, 11.0]]), torch.tensor([0.3, 0.9]), 0.5)
    for n in (50, 333, 2000):
        for thr in (0.1, 0.5):
            s, c = rand_segs(n, span=40.0 * n ** 0.5)
            add(f'rand{n}', s, c, thr)
    # dense cluster: everything overlaps, strong decay, pruning by min_score
    s, c = rand_segs(400, span=30.0, max_len=40.0)
    add('dense', s, c, 0.3, sigma=0.2, min_score=0.05)
    # nested chain: [0,100] > [1,99] > ... alternating scores
    k = 40
    chain = torch.stack((torch.arange(k, dtype=torch.float32), 100.0 - torch.arange(k, dtype=torch.float32)), -1)
    cs = torch.linspace(0.95, 0.05, k)[torch.randperm(k, generator=g)].contiguous()
    add('nested_chain', chain.contiguous(), cs, 0.7)
    # IoU exactly at threshold: [0,2] vs [1,3]: inter 1, union 3 -> 1/3; [0,4] vs [0,2]: 0.5
    eq = torch.tensor([[0.0, 4.0], [0.0, 2.0], [10.0, 12.0], [11.0, 13.0], [20.0, 21.0]])
    es = torch.tensor([0.9, 0.8, 0.7, 0.6, 0.5])
    add('iou_equal_half', eq, es, 0.5)
    add('iou_below', eq, es, 0.500001)
    out['cases'] = cases
    save('nms_kat.npz', out)


@torch.no_grad()
def gen_nms_big(ext):
    """more candidates than one workgroup's LDS holds (n > 4096): the reference extension takes any n (nms_cpu.cpp:20-63)"""
    g = torch.Generator().manual_seed(4321)
    out, cases = {}, []
    for n, thr, min_score in ((4097, 0.5, 0.001), (8192, 0.5, 0.001), (6000, 0.3, 0.05)):
        c = torch.rand(n, generator=g) * 60.0 * n ** 0.5
        ln = torch.rand(n, generator=g) * 60.0 + 0.2
        segs = torch.stack((c - ln / 2, c + ln / 2), -1).contiguous()
        scores = torch.rand(n, generator=g)
        while len(torch.unique(scores)) != n:
            scores = torch.rand(n, generator=g)
        i = len(cases)
        out[f'k{i}/segs'], out[f'k{i}/scores'] = segs, scores.contiguous()
        out[f'k{i}/nms'] = ext.nms(segs, scores, iou_thresh=float(thr))
        for method in (1, 2):
            dets = torch.full((n, 3), -7.0)
            idx = ext.softnms(segs, scores, dets, iou_thresh=float(thr), sigma=0.9, min_score=float(min_score), method=method)
            out[f'k{i}/soft{method}/idx'] = idx
            out[f'k{i}/soft{method}/dets'] = dets[:len(idx)]
        cases.append(dict(n=n, iou_thresh=thr, sigma=0.9, min_score=min_score))
    out['cases'] = cases
    save('nms_kat_big.npz', out)


# ------------------------------------------------------------------ G6: feature files / annotations / text-CLS table
def write_data_tree(root, arrays):
    """materialise the synthetic dataset of data_io.npz under ``root`` (also used by tests/test_data_io.py)"""
    import pickle
    for d in ('featA', 'featB', 'shallow', 'text', 'ext'):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    for vid in ('v0', 'v1'):
        a, b, sh = arrays[f'{vid}/featA'], arrays[f'{vid}/featB'], arrays[f'{vid}/shallow']
        np.save(os.path.join(root, 'featA', vid + '.npy'), a)
        np.save(os.path.join(root, 'featB', vid + '.npy'), b)
        np.save(os.path.join(root, 'shallow', vid + '.npy'), sh)
        torch.save(torch.from_numpy(np.array(a)), os.path.join(root, 'featA', vid + '.pt'))
        with open(os.path.join(root, 'featA', vid + '.pk'), 'wb') as fh:
            pickle.dump((np.array(a), np.array(a) * 0.5 + 1.0, np.array(a) * 0), fh)
    anno = json.loads(bytes(arrays['anno_json']).decode())
    with open(os.path.join(root, 'anno.json'), 'w') as fh:
        json.dump(anno, fh)
    cls = {}
    for i, sent in enumerate(json.loads(bytes(arrays['sentences']).decode())):
        cls[sent] = np.array(arrays['cls_rows'][i:i + 1])
        np.save(os.path.join(root, 'text', f't{i}.npy'), arrays[f'text/t{i}'])
        np.save(os.path.join(root, 'ext', f't{i}.npy'), arrays[f'ext/t{i}'])
    np.save(os.path.join(root, 'cls_val.npy'), cls, allow_pickle=True)


@torch.no_grad()
def gen_data_io():
    """Synthetic feature files pushed through the reference's OWN loaders (VID_LOAD_FUNC, VideoCentricDataset._load_vid_feats /
    _load_text_feats / _load_ext_scores / _parse_annotations / _load_text_cls_feats, libs/data/dataset.py) bound to a bare
    namespace instead of a constructed dataset (whose constructor wants the Ego4D metadata)."""
    import tempfile
    from types import SimpleNamespace as NS
    from libs.data import dataset as D
    g = np.random.default_rng(7)
    arrays = {}
    lens = {'v0': (37, 35), 'v1': (20, 20)}
    for vid, (la, lb) in lens.items():
        arrays[f'{vid}/featA'] = g.standard_normal((la, 6)).astype(np.float32)
        arrays[f'{vid}/featB'] = g.standard_normal((lb, 4)).astype(np.float32)
        arrays[f'{vid}/shallow'] = g.standard_normal(((la + 1) // 2, 10)).astype(np.float32)
    sentences = ['where is the red cup', 'who opened the door ', 'when did I wash hands', 'what fell down', 'zero length query']
    anno = {'val': {
        'v0': {'fps': 30.0, 'num_frames': 1200, 'num_clips': 37, 'annotations': [
            {'segment': [1.5, 7.25], 'sentence': sentences[0], 'sentence_id': 't0'},
            {'segment': [-2.0, 3.0], 'sentence': sentences[1], 'sentence_id': 't1'},
            {'segment': [38.0, 45.0], 'sentence': sentences[2], 'sentence_id': 't2'}]},
        'v1': {'fps': 25.0, 'num_frames': 500, 'duration': 19.5, 'annotations': [
            {'segment': [3.0, 9.0], 'sentence': sentences[3], 'sentence_id': 't3'},
            {'segment': [25.0, 30.0], 'sentence': sentences[4], 'sentence_id': 't4'}]},
        'no_queries': {'fps': 30.0, 'num_frames': 30}}}
    arrays['anno_json'] = np.frombuffer(json.dumps(anno).encode(), dtype=np.uint8)
    arrays['sentences'] = np.frombuffer(json.dumps(sentences).encode(), dtype=np.uint8)
    arrays['cls_rows'] = g.standard_normal((len(sentences), 10)).astype(np.float32)
    for i in range(len(sentences)):
        arrays[f'text/t{i}'] = g.standard_normal((5 + i, 8)).astype(np.float32)
        arrays[f'ext/t{i}'] = g.standard_normal((37,)).astype(np.float32)
    out = dict(arrays)
    with tempfile.TemporaryDirectory() as root:
        write_data_tree(root, arrays)
        for fmt in ('npy', 'pt', 'pk0', 'pk1', 'pk_avg'):
            out[f'clip/{fmt}'] = np.asarray(D.VID_LOAD_FUNC[fmt](os.path.join(root, 'featA', 'v0'), None))
        for tag, ds, norm in (('ds1', 1, False), ('ds2', 2, False), ('ds2_norm', 2, True)):
            for vid in lens:
                ns = NS(vid_feat_dict={}, vid_feat_dir=[os.path.join(root, 'featA'), os.path.join(root, 'featB')],
                        opt=NS(data=NS(vid_load='npy')), downsample_rate=ds, normalize_vid=norm)
                out[f'vid/{tag}/{vid}'] = D.VideoCentricDataset._load_vid_feats(ns, vid)
        ns = NS(text_feat_dict={}, tokenizer=None, text_feat_dir=os.path.join(root, 'text'), is_training=False, normalize_text=True)
        out['text_out_norm/t1'] = D.VideoCentricDataset._load_text_feats(ns, 't1')
        ns.normalize_text = False
        out['text_out/t1'] = D.VideoCentricDataset._load_text_feats(ns, 't1')
        ns = NS(ext_score_dir=os.path.join(root, 'ext'), downsample_rate=2, normalize_scores=True, temperature=0.7, text_feat_dict={})
        out['ext_norm/t0'] = D.VideoCentricDataset._load_ext_scores(ns, 't0')
        ns = NS(anno_file=os.path.join(root, 'anno.json'), split=('val',), opt=NS(data=NS(downsample_rate=2)))
        vid_dict, _ = D.VideoCentricDataset._parse_annotations(ns)
        out['parsed'] = {k: dict(fps=v['fps'], num_frames=v['num_frames'], num_clips=v['num_clips'], duration=v['duration'],
                                 text_ids=list(v['text_ids']), segments=np.asarray(v['segments']).tolist()) for k, v in vid_dict.items()}
        ns = NS(vid_dict=vid_dict, text_cls_dict=np.load(os.path.join(root, 'cls_val.npy'), allow_pickle=True).item())
        for vid in vid_dict:
            out[f'cls/{vid}'] = D.VideoCentricDataset._load_text_cls_feats(ns, vid, tuple(range(len(vid_dict[vid]['segments']))))
    save('data_io.npz', out)


if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ext = install_stubs()
    which = sys.argv[1:] or ['ops', 'gate', 'e2e', 'postproc', 'nms', 'data_io']
    if 'ops' in which:
        gen_ops()
    if 'ops64' in which or 'ops' in which:
        gen_ops64()
    if 'ops256' in which:
        gen_ops256()
    if 'gate' in which:
        gen_gate()
    if 'e2e' in which:
        gen_e2e()
    if 'variants' in which or 'e2e' in which:
        gen_e2e_variants()
    if 'text_identity' in which or 'e2e' in which:
        gen_text_identity()
    if 'scale' in which:
        gen_e2e_scale()
    if 'train' in which:
        gen_train()
    if 'train2' in which:
        gen_train_secondary()
    if 'postproc' in which:
        gen_postproc()
    if 'postproc_ext' in which or 'postproc' in which:
        gen_postproc_ext()
    if 'nms' in which:
        gen_nms(ext)
    if 'nms_big' in which:
        gen_nms_big(ext)
    if 'data_io' in which:
        gen_data_io()
