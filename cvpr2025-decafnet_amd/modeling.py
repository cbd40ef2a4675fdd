"""Host-side mirror of the reference ``libs/modeling`` interface for the grounding hot path.

``PtTransformerEarlyFusionIterative`` keeps the reference's constructor, method names,
argument meaning, return structure and -- above all -- its state_dict (the parameter ABI,
SURVEY.md 8b), so ``Evaluator`` (libs/worker_v2.py:726-1187) can use it unchanged.  The
modules below are parameter containers only: the eval forward is executed by the HIP
engine behind the C ABI (include/decafnet_hip.h); there is no PyTorch implementation of
the video path to fall back to.  The text encoder (<= 33 tokens per query, microseconds)
stays host-side PyTorch, as SURVEY.md 2 (#7) prescribes.

Reference citations are relative to the reference repository.
"""
from __future__ import annotations

import copy
import ctypes
import math
from typing import List, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib

__all__ = ['PtTransformerEarlyFusionIterative', 'PtTransformer', 'PtGenerator', 'create_model', 'sinusoid_encoding']


# ------------------------------------------------------------------------------------------
# parameter containers (names and shapes == reference)
# ------------------------------------------------------------------------------------------
class MaskedConv1D(nn.Module):
    """Holds ``conv`` like libs/modeling/blocks.py:63-85 (bias zero-initialised)."""

    def __init__(self, cin, cout, k, stride=1, padding=0, groups=1, bias=True):
        super().__init__()
        self.stride = stride
        self.conv = nn.Conv1d(cin, cout, k, stride=stride, padding=padding, groups=groups, bias=bias)
        if bias:
            nn.init.zeros_(self.conv.bias)


class LayerNorm(nn.Module):
    """Channel LayerNorm parameters, shaped (C, 1) as in blocks.py:119-121."""

    def __init__(self, c, affine=True):
        super().__init__()
        if affine:
            self.weight = nn.Parameter(torch.ones(c, 1))
            self.bias = nn.Parameter(torch.zeros(c, 1))
        else:
            self.weight = self.bias = None


class LayerScale(nn.Module):
    """blocks.py:670-678: per-channel residual scale (1, C, 1), init 1e-4."""

    def __init__(self, c, init_scale=1e-4):
        super().__init__()
        self.scale = nn.Parameter(init_scale * torch.ones((1, c, 1)))


class Scale(nn.Module):
    """blocks.py:653-664."""

    def __init__(self, init=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.as_tensor(init, dtype=torch.float))


class MaskedMHA(nn.Module):
    """Projection weights of blocks.py:145-200."""

    def __init__(self, embd_dim, q_dim=None, kv_dim=None, out_dim=None, n_heads=4, window_size=0):
        super().__init__()
        assert embd_dim % n_heads == 0
        q_dim = q_dim or embd_dim
        kv_dim = kv_dim or embd_dim
        out_dim = out_dim or q_dim
        self.n_heads, self.window_size = n_heads, window_size
        self.query = nn.Conv1d(q_dim, embd_dim, 1)
        self.key = nn.Conv1d(kv_dim, embd_dim, 1)
        self.value = nn.Conv1d(kv_dim, embd_dim, 1)
        self.proj = nn.Conv1d(embd_dim, out_dim, 1)


class ConvAttNLayer(nn.Module):
    """blocks.py:414-460."""

    def __init__(self, embd_dim, stride=1, n_heads=4, window_size=0):
        super().__init__()
        if stride > 0:
            for n in 'qkv':
                setattr(self, f'{n}_conv', MaskedConv1D(embd_dim, embd_dim, 3, stride, 1, groups=embd_dim, bias=False))
            for n in 'qkv':
                setattr(self, f'{n}_norm', LayerNorm(embd_dim))
        self.attn = MaskedMHA(embd_dim, n_heads=n_heads, window_size=window_size)


class FFN(nn.Module):
    """blocks.py:523-533."""

    def __init__(self, c, expansion=4):
        super().__init__()
        self.fc = nn.Conv1d(c, c * expansion, 1)
        self.proj = nn.Conv1d(c * expansion, c, 1)


class TransformerEncoder(nn.Module):
    """blocks.py:541-576."""

    def __init__(self, embd_dim, stride=1, n_heads=4, window_size=0):
        super().__init__()
        self.stride, self.n_heads, self.window_size = stride, n_heads, window_size
        self.attn = ConvAttNLayer(embd_dim, stride, n_heads, window_size)
        self.ln_attn = LayerNorm(embd_dim)
        self.drop_path_attn = LayerScale(embd_dim)
        self.ffn = FFN(embd_dim)
        self.ln_ffn = LayerNorm(embd_dim)
        self.drop_path_ffn = LayerScale(embd_dim)


class ConvXAttNLayer(nn.Module):
    """blocks.py:476-511."""

    def __init__(self, embd_dim, kv_dim, out_dim, n_heads=4):
        super().__init__()
        self.q_conv = MaskedConv1D(embd_dim, embd_dim, 3, 1, 1, groups=embd_dim, bias=False)
        self.q_norm = LayerNorm(embd_dim)
        self.xattn = MaskedMHA(embd_dim, kv_dim=kv_dim, out_dim=out_dim, n_heads=n_heads)


class TransformerDecoder(nn.Module):
    """blocks.py:594-630; xattn_mode 'adaln' (the cross-attention output modulates LayerNorm(q)) or 'affine' (it modulates q
    itself).  Neither mode has parameters of its own: the adaln LayerNorm is affine=False (blocks.py:623-626)."""

    def __init__(self, embd_dim, kv_dim, n_heads=4, xattn_mode='adaln'):
        super().__init__()
        assert xattn_mode in ('affine', 'adaln')                         # blocks.py:613
        self.xattn_mode = xattn_mode
        self.xattn = ConvXAttNLayer(embd_dim, kv_dim, embd_dim * 2, n_heads)
        self.ln_xattn_q = LayerNorm(embd_dim)
        self.ln_xattn_kv = LayerNorm(kv_dim)
        self.ffn = FFN(embd_dim)
        self.ln_ffn = LayerNorm(embd_dim)
        self.drop_path_ffn = LayerScale(embd_dim)


class XAttNFusion(nn.Module):
    """fusion.py:16-54."""

    def __init__(self, vid_dim, text_dim, n_layers=2, n_heads=4, xattn_mode='adaln', **_):
        super().__init__()
        self.n_heads = n_heads
        self.layers = nn.ModuleList(TransformerDecoder(vid_dim, text_dim, n_heads, xattn_mode) for _ in range(n_layers))
        self.ln_out = LayerNorm(vid_dim)


class VideoTransformer(nn.Module):
    """video_net.py:20-121.  ``stride`` = s > 1: the first log2(s) embedding convolutions are k5 / stride 2 / padding 2
    (:59-74); ``pool_only``: a branch layer is one depthwise k3 MaskedConv1D (:98-111)."""

    def __init__(self, in_dim, embd_dim, max_seq_len, n_heads, mha_win_size, stride=1, arch=(2, 1, 6),
                 use_abs_pe=False, pool_only=False, **_):
        super().__init__()
        assert len(arch) == 3
        assert stride >= 1 and stride & (stride - 1) == 0 and arch[0] >= int(math.log2(stride))      # video_net.py:52-53
        self.max_seq_len, self.embd_dim, self.arch = max_seq_len, embd_dim, tuple(arch)
        self.n_heads, self.mha_win_size, self.use_abs_pe = n_heads, mha_win_size, use_abs_pe
        self.stride, self.pool_only = int(stride), bool(pool_only)
        self.embd_fc = MaskedConv1D(in_dim, embd_dim, 1)
        convs, s = [], int(stride)
        for _i in range(arch[0]):
            convs.append(MaskedConv1D(embd_dim, embd_dim, 5 if s > 1 else 3, 2 if s > 1 else 1, 2 if s > 1 else 1, bias=False))
            s = max(s // 2, 1)
        self.embd_convs = nn.ModuleList(convs)
        self.embd_norms = nn.ModuleList(LayerNorm(embd_dim) for _ in range(arch[0]))
        self.stem = nn.ModuleList(TransformerEncoder(embd_dim, 1, n_heads, mha_win_size) for _ in range(arch[1]))
        if pool_only:
            self.branch = nn.ModuleList(MaskedConv1D(embd_dim, embd_dim, 3, 2 if i > 0 else 1, 1, groups=embd_dim, bias=False)
                                        for i in range(arch[2]))
        else:
            self.branch = nn.ModuleList(TransformerEncoder(embd_dim, 2 if i > 0 else 1, n_heads, mha_win_size)
                                        for i in range(arch[2]))
        for mod in self.modules():
            if isinstance(mod, nn.Conv1d) and mod.bias is not None:
                nn.init.zeros_(mod.bias)


class ConvHead(nn.Module):
    """ClsHead / RegHead parameters, head.py:18-51 and :67-93."""

    def __init__(self, embd_dim, out_name, out_dim, n_layers=2, num_fpn_levels=None, prior_prob=0.0):
        super().__init__()
        self.convs = nn.ModuleList(MaskedConv1D(embd_dim, embd_dim, 3, 1, 1, bias=False) for _ in range(n_layers))
        self.norms = nn.ModuleList(LayerNorm(embd_dim) for _ in range(n_layers))
        setattr(self, out_name, MaskedConv1D(embd_dim, out_dim, 3, 1, 1))
        if out_name == 'cls_head' and prior_prob > 0:
            nn.init.constant_(self.cls_head.conv.bias, -math.log((1 - prior_prob) / prior_prob))
        if num_fpn_levels is not None:
            self.scales = nn.ModuleList(Scale() for _ in range(num_fpn_levels))


class DilatedResidualLayer(nn.Module):
    """tcn.py:4-19."""

    def __init__(self, dilation, c):
        super().__init__()
        self.conv_dilated = nn.Conv1d(c, c, 3, padding=dilation, dilation=dilation)
        self.conv_1x1 = nn.Conv1d(c, c, 1)
        self.norm = nn.LayerNorm(c, eps=1e-5)


class TCN(nn.Module):
    """tcn.py:40-58 with in_map=True."""

    def __init__(self, in_dim, hid_dim, out_dim, num_layers):
        super().__init__()
        self.conv_1x1 = nn.Conv1d(in_dim, hid_dim, 1)
        self.layers = nn.ModuleList(DilatedResidualLayer(2 ** i, hid_dim) for i in range(num_layers))
        self.conv_out = nn.Conv1d(hid_dim, out_dim, 1)


def sinusoid_encoding(seq_len, n_freqs):
    """blocks.py:134-142."""
    tics = torch.arange(seq_len, dtype=torch.float)
    freqs = 10000 ** torch.linspace(0, 1, n_freqs + 1)[:n_freqs]
    x = tics[None, :] / freqs[:, None]
    return torch.cat((torch.sin(x), torch.cos(x)))


# ------------------------------------------------------------------------------------------
# text encoder (host-side PyTorch; runs on whatever device its parameters live on)
# ------------------------------------------------------------------------------------------
def _chan_ln(x, ln: LayerNorm, eps=1e-5):
    x = x - x.mean(dim=1, keepdim=True)
    x = x / torch.sqrt((x * x).mean(dim=1, keepdim=True) + eps)
    return x * ln.weight + ln.bias


class TextTransformer(nn.Module):
    """text_net.py:92-188."""

    def __init__(self, in_dim, embd_dim, n_heads, max_seq_len, n_layers=5, use_abs_pe=True, use_bkgd_token=True, **_):
        super().__init__()
        self.max_seq_len, self.n_heads, self.in_dim, self.embd_dim = max_seq_len, n_heads, in_dim, embd_dim
        self.use_abs_pe, self.use_bkgd_token = bool(use_abs_pe), bool(use_bkgd_token)
        self.embd_fc = MaskedConv1D(in_dim, embd_dim, 1)
        if use_abs_pe:
            pe = sinusoid_encoding(max_seq_len, embd_dim // 2) / embd_dim ** 0.5
            self.register_buffer('pe', pe, persistent=False)
        else:
            self.pe = None
        if use_bkgd_token:
            self.bkgd_token = nn.Parameter(torch.empty(embd_dim, 1))
            nn.init.trunc_normal_(self.bkgd_token, mean=0.0, std=0.02, a=-2.0, b=2.0)
        else:
            self.bkgd_token = None
        self.transformer = nn.ModuleList(TransformerEncoder(embd_dim, 0, n_heads, 0) for _ in range(n_layers))
        for mod in self.modules():
            if isinstance(mod, nn.Conv1d) and mod.bias is not None:
                nn.init.zeros_(mod.bias)

    def forward(self, x, mask):
        raise RuntimeError('TextTransformer is a parameter container here: call model.encode_text(tokens, token_masks), '
                           'which runs the text encoder on the MI355X (dcf_text_encode); there is no CPU path')


class _MHAParams(nn.Module):
    """parameter names of MaskedMHA (blocks.py:166-201): query / key / value / proj 1x1 convolutions"""

    def __init__(self, embd_dim):
        super().__init__()
        self.query = nn.Conv1d(embd_dim, embd_dim, 1)
        self.key = nn.Conv1d(embd_dim, embd_dim, 1)
        self.value = nn.Conv1d(embd_dim, embd_dim, 1)
        self.proj = nn.Conv1d(embd_dim, embd_dim, 1)


class _AttNPool1D(nn.Module):
    """AttNPool1D (blocks.py:396-411): parameter container (`attn_pool.attn.*`)"""

    def __init__(self, embd_dim, n_heads):
        super().__init__()
        self.attn = _MHAParams(embd_dim)
        self.n_heads = n_heads


class TextIdentity(nn.Module):
    """text_net.py:22-89 (`opt.model.text_net.name == 'identity'`): optional 1x1 embedding, optional position encoding,
    an attention-pooled summary token in place of the learned background token.  Parameter container: the forward runs
    on the MI355X (dcf_text_encode, dcf_config.text_kind = 1)."""

    def __init__(self, in_dim, embd_dim, max_seq_len, n_heads=4, use_abs_pe=False, use_bkgd_token=True, **_):
        super().__init__()
        self.max_seq_len, self.n_heads, self.in_dim = max_seq_len, n_heads, in_dim
        self.embd_fc = MaskedConv1D(in_dim, embd_dim, 1) if embd_dim is not None else None
        self.embd_dim = embd_dim if embd_dim is not None else in_dim
        self.use_abs_pe, self.use_bkgd_token = bool(use_abs_pe), bool(use_bkgd_token)
        if use_abs_pe:
            pe = sinusoid_encoding(max_seq_len, self.embd_dim // 2) / self.embd_dim ** 0.5
            self.register_buffer('pe', pe, persistent=False)
        else:
            self.pe = None
        self.attn_pool = _AttNPool1D(self.embd_dim, n_heads) if use_bkgd_token else None
        self.transformer = ()
        for mod in self.modules():
            if isinstance(mod, nn.Conv1d) and mod.bias is not None:
                nn.init.zeros_(mod.bias)

    def forward(self, x, mask):
        raise RuntimeError('TextIdentity is a parameter container here: call model.encode_text(tokens, token_masks), '
                           'which runs the text path on the MI355X (dcf_text_encode); there is no CPU path')


def make_text_net(tn):
    """make_text_net (text_net.py:191-193): 'transformer' or 'identity'"""
    tn = dict(tn)
    name = tn.pop('name', 'transformer')
    if name == 'transformer':
        return TextTransformer(**tn)
    if name == 'identity':
        tn.pop('n_layers', None)
        return TextIdentity(**tn)
    raise NotImplementedError(f'text_net.name = {name!r}: only the transformer and identity text backbones are supported')


# ------------------------------------------------------------------------------------------
# the model
# ------------------------------------------------------------------------------------------
class _Engine:
    """Owns one dcf_model handle and keeps it bound to the module's current parameter storage."""

    def __init__(self, cfg: _lib.DcfConfig):
        self.lib = _lib.lib()
        self.handle = ctypes.c_void_p()
        _lib.check(self.lib.dcf_model_create(ctypes.byref(cfg), ctypes.byref(self.handle)), 'dcf_model_create')
        self.signature = None
        self.cached = None
        self.cached_ids = None
        self.keepalive = []
        self.status_probe = None                     # (pinned int32, event) of the last forward's numerics word
        self.pe_cache = {}
        self.text_pe_cache = {}

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.dcf_model_destroy(self.handle)
        except Exception:
            pass

    def set_graph_mode(self, mode):
        code = {'auto': 0, 'always': 1, 'never': 2}[mode]
        if getattr(self, '_graph_mode', None) != code:
            _lib.check(self.lib.dcf_model_set_graph_mode(self.handle, code), 'dcf_model_set_graph_mode')
            self._graph_mode = code

    def set_ln_carry(self, on):
        if getattr(self, '_ln_carry', True) != bool(on):
            _lib.check(self.lib.dcf_model_set_ln_carry(self.handle, int(bool(on))), 'dcf_model_set_ln_carry')
            self._ln_carry = bool(on)

    def bind(self, model):
        """(Re)bind the module's parameters if their storage or contents changed.  The walk over the module tree
        (state_dict) costs about as much host time as a whole T = 16384 forward takes on the GPU, so the tensor list is
        cached per engine and only a cheap signature -- storage address and in-place version counter of every cached
        parameter -- is compared on the hot path (``.cuda()`` / ``.to()`` swap ``param.data``, ``load_state_dict`` copies
        in place and bumps ``_version``; both keep the Parameter objects)."""
        # replaced Parameter OBJECTS (module.weight = nn.Parameter(...), load_state_dict(assign=True), re-parametrisation)
        # keep neither the storage nor the version counter of the cached ones: the identity of every parameter slot is
        # part of the check (a walk over _parameters dicts, no state_dict construction: ~0.1 ms for 400 tensors)
        self.set_ln_carry(getattr(model, 'ln_carry', True))
        ids = model._parameter_ids()
        if self.cached is None or ids != self.cached_ids:
            self.cached = model._named_engine_tensors()
            self.cached_ids = ids
            self.signature = None
        named_tensors = self.cached
        self.set_graph_mode(getattr(model, 'graph_mode', 'auto'))
        sig = tuple((t.data_ptr(), t._version) for _, t in named_tensors)
        if sig == self.signature:
            return
        keep = []
        for name, t in named_tensors:
            if not (t.is_cuda and t.dtype == torch.float32):
                raise RuntimeError(f'parameter {name} must be a float32 tensor on the GPU (got {t.dtype} on {t.device}); '
                                   f'call model.cuda() first -- there is no CPU path')
            tc = t.detach().contiguous()
            keep.append(tc)
            shape = (ctypes.c_int64 * max(tc.dim(), 1))(*(tc.shape if tc.dim() else (1,)))
            _lib.check(self.lib.dcf_model_bind(self.handle, name.encode(), _lib.ptr(tc), shape, max(tc.dim(), 1)),
                       f'dcf_model_bind({name})')
        _lib.check(self.lib.dcf_model_finalize(self.handle, _lib.current_stream()), 'dcf_model_finalize')
        self.keepalive = keep
        self.signature = sig


def _text_config(c, tn):
    c.text_in, c.text_layers, c.text_heads = tn.in_dim, len(tn.transformer), tn.n_heads
    c.text_abs_pe, c.text_bkgd = int(tn.use_abs_pe), int(tn.use_bkgd_token)
    c.text_kind = int(isinstance(tn, TextIdentity))


def _encode_text(model, tokens, token_masks):
    """TextTransformer.forward through the C ABI, one query at a time (the reference calls it with bs = 1,
    worker_v2.py:953)."""
    if not tokens.is_cuda:
        raise RuntimeError('encode_text runs on the MI355X only: move the tokens to the GPU')
    if model._engine is None:
        model._engine = _Engine(model._config())
    model._raise_if_flagged()
    eng = model._engine
    eng.bind(model)
    tn = model.text_net
    bs, ct, lq = tokens.shape
    assert ct == tn.in_dim, (ct, tn.in_dim)
    lk = lq + int(tn.use_bkgd_token)
    te = tn.embd_dim
    if token_masks is None:
        token_masks = torch.ones(bs, 1, lq, dtype=torch.bool, device=tokens.device)
    masks = token_masks.reshape(bs, lq).to(torch.bool).contiguous()
    if tn.use_abs_pe:
        key = lq
        if key not in eng.text_pe_cache:
            pe = tn.pe.float()
            if lq > tn.max_seq_len:                                              # text_net.py:172-177
                pe = F.interpolate(pe[None], size=lq, mode='linear', align_corners=True)[0]
            eng.text_pe_cache[key] = pe[:, :lq].t().contiguous().to(tokens.device)
        pe_t = eng.text_pe_cache[key]
        _lib.check(eng.lib.dcf_model_set_text_pe(eng.handle, _lib.ptr(pe_t), lq), 'dcf_model_set_text_pe')
    out = torch.empty(bs, te, lk, dtype=torch.float32, device=tokens.device)
    out_mask = torch.empty(bs, 1, lk, dtype=torch.bool, device=tokens.device)
    tok = tokens.contiguous().float()
    for b in range(bs):
        _lib.check(eng.lib.dcf_text_encode(eng.handle, _lib.ptr(tok[b]), _lib.ptr(masks[b]), lq, _lib.ptr(out[b]),
                                           _lib.ptr(out_mask[b]), _lib.current_stream()), 'dcf_text_encode')
    return out, out_mask


GEMM_MODES = {'f16x3': 16, 'bf16x6': 6, 'fp32': 1}
# arithmetic of the attention products on the matrix cores (extension key opt.model.attn_mode): 'f16x3' (default, fp32 accurate) or
# 'f16': one fp16 product per multiply-add (opt-in, ~1e-4 on the logits; BASELINE configs[4]'s "bf16 MFMA attention" with fp16's
# three extra bits)
ATTN_MODES = {'f16x3': 0, 'f16': 1}


class PtTransformerEarlyFusionIterative(nn.Module):
    """Drop-in for libs/modeling/model.py:397-565 (created by libs/worker_v2.py:182-211).

    ``opt`` needs ``opt.model.{vid_net,text_net,fusion,cls_head,reg_head,sn,sratio,msf,scat,sfonly,norm}``
    (attribute or item access).  Unlike the reference the constructor does not mutate ``opt``.
    """

    MODEL_KIND = 0

    def __init__(self, opt, second_fusion=True):
        super().__init__()
        mo = opt['model'] if isinstance(opt, dict) else opt.model
        mo = copy.deepcopy(mo)
        self.opt = opt
        vn, tn, fu = dict(mo['vid_net']), dict(mo['text_net']), dict(mo['fusion'])
        if vn.get('name', 'transformer') != 'transformer':
            raise NotImplementedError('only the transformer video backbone is supported')
        self.sn, self.sratio = int(mo['sn']), float(mo['sratio'])
        self.msf, self.norm = bool(mo['msf']), bool(mo['norm'])
        self.scat, self.sfonly = bool(mo.get('scat', False)), bool(mo.get('sfonly', False))
        in_dim, E = int(vn['in_dim']), int(vn['embd_dim'])
        # model.py:409-416: vid_map takes in_dim (x2 with msf) (+1 with scat) channels.  With msf and sfonly its input is
        # the sidekick features alone (model.py:546-547), so the feature files are 2*in_dim wide in that configuration.
        map_in = (2 * in_dim if self.msf else in_dim)
        D = map_in if (self.msf and self.sfonly) else in_dim
        self.D, self.E = D, E

        self.text_net = make_text_net(tn)
        self.vid_map = MaskedConv1D(map_in + int(self.scat), E, 1)
        vn.pop('name', None)
        vn['in_dim'] = E
        self.vid_net = VideoTransformer(**vn)
        fu.pop('name', None)
        self.fusion = XAttNFusion(**fu)
        ch, rh = dict(mo['cls_head']), dict(mo['reg_head'])
        n_levels = self.vid_net.arch[2]
        self.cls_head = ConvHead(ch['embd_dim'], 'cls_head', 1, ch.get('n_layers', 2), None, ch.get('prior_prob', 0.0))
        self.refine = TCN(n_levels, 32, 32, num_layers=n_levels)
        self.cls_head2 = ConvHead(ch['embd_dim'] + 32, 'cls_head', 1, ch.get('n_layers', 2), None, ch.get('prior_prob', 0.0))
        self.reg_head = ConvHead(rh['embd_dim'] + 32, 'reg_head', 2, rh.get('n_layers', 2),
                                 rh.get('num_fpn_levels', n_levels))
        self.second_fusion = second_fusion
        self.head_layers = ch.get('n_layers', 2)
        self.max_batch = int(mo.get('max_batch', 0) or 0)
        # dense-conv arithmetic (extension key opt.model.gemm_mode), all fp32 accurate: 'f16x3' (default: two fp16 planes
        # per operand on the fp16 matrix cores), 'bf16x6' (three bf16 planes) or 'fp32' (native fp32 MFMA)
        self.gemm_mode = GEMM_MODES[mo.get('gemm_mode', 'f16x3')]
        self.attn_mode = ATTN_MODES[mo.get('attn_mode', 'f16x3')]
        self._engine = None
        # Serving option (off by default: the reference returns fresh tensors every call).  When True the three flat
        # output buffers are reused between calls of the same shape, which lets the engine replay one captured HIP
        # graph instead of ~135 kernel launches -- results of call k are overwritten by call k+1.
        self.reuse_output_buffers = False
        self.ln_carry = True                        # LayerNorms carried between kernels as row statistics where the kernels allow (set_ln_carry)
        self._out_cache = {}
        # 'auto' (replay a HIP graph for large batched forwards, launch small ones eagerly: dcf_model_set_graph_mode),
        # 'always' or 'never'
        self.graph_mode = 'auto'

    # -- reference API ---------------------------------------------------------------------
    def encode_text(self, tokens, token_masks):
        """model.py:434-436: tokens (bs, C_t, Lq) f32, token_masks (bs, 1, Lq) bool -> ((bs, TE, Lk), (bs, 1, Lk)),
        Lk = Lq + use_bkgd_token.  Runs TextTransformer.forward (text_net.py:158-188) on the GPU (dcf_text_encode)."""
        return _encode_text(self, tokens, token_masks)

    def forward(self, vid, shallow_vid, vid_masks, text, text_cls, text_masks, text_size=None, mv_data=None, eval=False):
        if not eval:
            return self._drop_forward(vid, shallow_vid, vid_masks, text, text_cls, text_masks, text_size, mv_data, eval)
        return self._drop_forward_eval(vid, shallow_vid, vid_masks, text, text_cls, text_masks, text_size, mv_data, eval)

    def _drop_forward(self, vid, shallow_vid, vid_masks, text, text_cls, text_masks, text_size=None, mv_data=None, eval=False):
        """Training-mode forward, FORWARD VALUES ONLY (model.py:567-632): vid / shallow_vid (bs, D, T), vid_masks (bs, T),
        raw text tokens (sum(text_size), C_t, Lq) -- or padded (bs, max_k, C_t, Lq), model.py:617-622 -- with text_masks
        (.., 1, Lq) / (.., Lq), text_cls (sum(text_size), D), text_size (bs,) queries per video (None: one per video).
        Video b is repeated for its text_size[b] queries (model.py:579-582), the text encoder runs inside (model.py:624).
        Returns what the reference returns: (fpn_logits1, fpn_logits2, fpn_offsets, fpn_masks), tuples over the L levels of
        (B', T_l) / (B', T_l) / (B', T_l, 2) / (B', T_l) bool with B' = sum(text_size) rows in (video, query) order; the classes with
        one classification head (PtTransformer, PtTransformerEarlyFusion: model.py:110-161, :300-373) return (fpn_logits, fpn_offsets,
        fpn_masks) as theirs do.
        There is no backward pass and no random number stream here: every dropout / drop-path probability of ``opt`` must
        be 0, and the Dropout(0.5) the reference hard-codes into every layer of the refinement TCN (tcn.py:5,13;
        model.py:424-425 passes no dropout argument) is taken at p = 0 too -- the values are those of the reference's
        train()-mode forward with that module's dropout disabled.  The outputs carry no autograd graph; the reference's Trainer
        is out of scope."""
        assert mv_data is None
        mo = self.opt['model'] if isinstance(self.opt, dict) else self.opt.model
        for part in ('vid_net', 'text_net', 'fusion'):
            for key in ('attn_pdrop', 'proj_pdrop', 'path_pdrop', 'cdrop'):
                if float(dict(mo[part]).get(key, 0.0) or 0.0) != 0.0:
                    raise NotImplementedError(f'training-mode forward: opt.model.{part}.{key} must be 0 (forward values only: no dropout '
                                              f'random stream, no backward pass)')
        if not vid.is_cuda:
            raise RuntimeError('the grounding forward runs on the MI355X only: move the inputs to the GPU')
        bs, T = vid.size(0), vid.size(-1)
        sizes = [1] * bs if text_size is None else [int(k) for k in text_size]
        assert len(sizes) == bs and all(k >= 1 for k in sizes)
        if text.ndim == 4:                                                    # padded per video, model.py:617-622
            text = torch.cat([t[:k] for t, k in zip(text, sizes)])
            if text_masks.ndim == 3:
                text_masks = torch.cat([t[:k] for t, k in zip(text_masks, sizes)])
        nq = sum(sizes)
        assert text.size(0) == nq and text_cls.size(0) == nq, (text.shape, text_cls.shape, sizes)
        enc, enc_mask = self.encode_text(text, text_masks.reshape(nq, 1, -1))  # (nq, TE, Lk), (nq, 1, Lk)
        dev = vid.device
        eng = self._engine
        lib = eng.lib
        keep = []
        vptr, sptr, mptr_v, cptr = ((ctypes.c_void_p * bs)() for _ in range(4))
        nqs = (ctypes.c_int32 * bs)()
        tptr, mptr, tlen = (ctypes.c_void_p * nq)(), (ctypes.c_void_p * nq)(), (ctypes.c_int32 * nq)()
        q = 0
        for b in range(bs):
            vc, sc = vid[b].contiguous().float(), shallow_vid[b].contiguous().float()
            mc = vid_masks[b].reshape(-1).to(torch.bool).contiguous()
            cc = text_cls[q:q + sizes[b]].contiguous().float()
            assert sc.shape == vc.shape == (self.D, T) and mc.numel() == T and cc.shape == (sizes[b], self.D)
            keep += [vc, sc, mc, cc]
            vptr[b], sptr[b], mptr_v[b], cptr[b], nqs[b] = vc.data_ptr(), sc.data_ptr(), mc.data_ptr(), cc.data_ptr(), sizes[b]
            for i in range(q, q + sizes[b]):
                t = enc[i].contiguous().float()
                m = enc_mask[i].reshape(-1).to(torch.bool).contiguous()
                keep += [t, m]
                tptr[i], mptr[i], tlen[i] = t.data_ptr(), m.data_ptr(), t.size(1)
            q += sizes[b]
        pe = None
        if self.vid_net.use_abs_pe:
            Tp = T // self.vid_net.stride            # the pyramid (and its position encoding) starts behind the strided convolutions
            pe = self._position_encoding(Tp, dev)
            _lib.check(lib.dcf_model_set_pe(eng.handle, _lib.ptr(pe), Tp), 'dcf_model_set_pe')
        S = lib.dcf_points_per_query(eng.handle, T)
        logits2 = torch.empty(nq, S, device=dev, dtype=torch.float32)
        offsets = torch.empty(nq, S, 2, device=dev, dtype=torch.float32)
        masks = torch.empty(nq, S, device=dev, dtype=torch.bool)
        sizes_l = [(T // self.vid_net.stride) >> l for l in range(self.vid_net.arch[2])]
        if self.MODEL_KIND != 0:
            # PtTransformer (model.py:110-161) / PtTransformerEarlyFusion (model.py:300-373): one classification head, no refinement
            # branch -- the training forward at p = 0 computes, per (video, query) row, exactly what the evaluation forward computes
            # (the gate, vid_map, fusion, encoder and heads of model.py:83-147 / :320-362 do not depend on `eval`), so it runs the
            # evaluation engine over the (video, query) rows and returns the reference's three tuples
            _lib.check(lib.dcf_forward_eval_videos(eng.handle, bs, vptr, sptr, mptr_v, T, nqs, tptr, mptr, tlen, cptr, _lib.ptr(logits2),
                                                   _lib.ptr(offsets), _lib.ptr(masks), _lib.current_stream()), 'dcf_forward_eval_videos')
            self._last_inputs = (keep, pe, enc, enc_mask)
            self._last_flat = (logits2, offsets, masks)
            self._probe_numerics()
            return tuple(logits2.split(sizes_l, 1)), tuple(offsets.split(sizes_l, 1)), tuple(masks.split(sizes_l, 1))
        logits1 = torch.empty(nq, S, device=dev, dtype=torch.float32)
        _lib.check(lib.dcf_forward_train_videos(eng.handle, bs, vptr, sptr, mptr_v, T, nqs, tptr, mptr, tlen, cptr, _lib.ptr(logits1),
                                                _lib.ptr(logits2), _lib.ptr(offsets), _lib.ptr(masks), _lib.current_stream()),
                   'dcf_forward_train_videos')
        self._last_inputs = (keep, pe, enc, enc_mask)
        self._last_flat = (logits2, offsets, masks)
        self._probe_numerics()
        return (tuple(logits1.split(sizes_l, 1)), tuple(logits2.split(sizes_l, 1)), tuple(offsets.split(sizes_l, 1)),
                tuple(masks.split(sizes_l, 1)))

    def numerics_status(self, reset=False):
        """dcf_numerics_status of the engine (blocking): bit 0 = a GEMM of the f16x3 mode produced a non-finite value since
        the last reset (an activation beyond |a| < 4094): re-run with opt.model.gemm_mode = 'bf16x6'; 16 = a LayerNorm carried as
        one-pass row statistics met an ill-conditioned row: set_ln_carry(False) and repeat."""
        if self._engine is None:
            return 0
        rc = self._engine.lib.dcf_numerics_status(self._engine.handle, int(bool(reset)), _lib.current_stream())
        if rc < 0:
            _lib.check(rc, 'dcf_numerics_status')
        return rc

    def set_ln_carry(self, on):
        """dcf_model_set_ln_carry: False = every LayerNorm of the model runs as its own two-pass launch instead of riding between
        kernels as one-pass row statistics (what numerics_status() & 16 asks for)"""
        self.ln_carry = bool(on)
        if self._engine is not None:
            self._engine.set_ln_carry(on)

    def _ln_carry_tripped(self):
        """numerics_status() & 16: a carried LayerNorm met a row whose mean dwarfs its spread (mean^2 > 64 var).  The model switches
        to the two-pass LayerNorm launches by itself; the forwards since the last check have to be repeated."""
        self.set_ln_carry(False)
        self.numerics_status(reset=True)
        raise RuntimeError("a LayerNorm carried between kernels as one-pass row statistics met a row whose mean dwarfs its spread "
                           "(|mean| > 8 sigma) in an earlier forward of this model: its variance lost accuracy to cancellation and the "
                           "logits of that forward were set to NaN.  The model now runs every LayerNorm as its own two-pass launch "
                           "(set_ln_carry(False)); repeat the forward(s) since the last check.  The flag has been reset.  "
                           "(GroundingEvaluator repeats them by itself.)")

    def ln_carry_flag_nowait(self):
        """True when the last completed numerics probe of this model (the 4-byte async copy every forward ends with) shows the one-pass
        LayerNorm guard raised.  Never waits: the word is whatever the newest finished copy left in pinned memory, so a caller who has
        synchronised with a forward's outputs sees that forward's flag (the word is sticky until reset)."""
        eng = self._engine
        return bool(eng is not None and eng.status_probe is not None and int(eng.status_probe[0][0]) & 2)

    def acknowledge_ln_carry(self):
        """The caller has seen numerics_status() & 16 (or ln_carry_flag_nowait()) and will REPEAT the affected forwards itself: switch to
        the two-pass LayerNorm launches, reset the sticky word and disarm the pending probe (no exception at the next call)."""
        self.set_ln_carry(False)
        self.numerics_status(reset=True)
        eng = self._engine
        if eng is not None and eng.status_probe is not None:
            eng.status_probe[0].zero_()
            eng.status_probe[2] = False

    def graph_active(self):
        """dcf_graph_active: how the last forward was issued (0 eager launches, 1 HIP-graph replay, 2 capture + launch)"""
        if self._engine is None:
            return 0
        return int(self._engine.lib.dcf_graph_active(self._engine.handle))

    def replica(self):
        """A second handle on the SAME parameters with its own engine (workspace, repacked weights, HIP graph): run it on
        another HIP stream to keep several videos in flight (throughput mode, see bench.py).  Parameters are shared by
        reference; call after ``.cuda().eval()``."""
        other = copy.copy(self)                     # shallow: _parameters / _modules / _buffers are the same objects
        other._engine = None
        other._out_cache = {}
        other._last_inputs = other._last_flat = None
        return other

    # -- HIP path ----------------------------------------------------------------------------
    def _config(self) -> _lib.DcfConfig:
        c = _lib.DcfConfig()
        vn = self.vid_net
        c.D, c.E, c.TE = self.D, self.E, self.fusion.layers[0].ln_xattn_kv.weight.shape[0] if len(self.fusion.layers) else self.E
        c.vid_heads, c.fusion_heads, c.fusion_layers = vn.n_heads, self.fusion.n_heads, len(self.fusion.layers)
        c.n_embd_convs, c.n_stem, c.n_levels = vn.arch
        c.win, c.head_layers = vn.mha_win_size, self.head_layers
        c.sn, c.sratio, c.msf, c.norm = self.sn, self.sratio, int(self.msf), int(self.norm)
        c.scat, c.sfonly = int(self.scat), int(self.sfonly and self.MODEL_KIND == 0)
        c.use_abs_pe, c.max_batch = int(vn.use_abs_pe), self.max_batch
        c.gemm_mode = self.gemm_mode
        c.model_kind, c.second_fusion = self.MODEL_KIND, int(bool(self.second_fusion))
        c.xattn_affine = int(len(self.fusion.layers) > 0 and self.fusion.layers[0].xattn_mode == 'affine')
        c.vid_stride, c.pool_only = vn.stride, int(vn.pool_only)
        c.attn_mode = getattr(self, 'attn_mode', 0)
        _text_config(c, self.text_net)
        return c

    def _named_engine_tensors(self):
        return list(self.state_dict(keep_vars=True).items())

    def _parameter_ids(self):
        return tuple(id(p) for mod in self.modules() for p in mod._parameters.values())

    # The f16x3 range flag of a plain model(...) call: the forward ends with a 4-byte async copy of the sticky word into
    # pinned memory; the NEXT call into the model (forward / forward_videos / encode_text) looks at it if that copy has
    # completed -- no wait -- and raises.  (The forward itself has already overwritten its logits with NaN on the device.)
    def _probe_numerics(self):
        eng = self._engine
        if eng.status_probe is None:
            eng.status_probe = [torch.zeros(1, dtype=torch.int32).pin_memory(), torch.cuda.Event(), False]
        host, ev, _ = eng.status_probe
        _lib.check(eng.lib.dcf_numerics_status_async(eng.handle, ctypes.c_void_p(host.data_ptr()), _lib.current_stream()),
                   'dcf_numerics_status_async')
        ev.record()
        eng.status_probe[2] = True

    def _raise_if_flagged(self):
        eng = self._engine
        if eng is None or eng.status_probe is None or not eng.status_probe[2]:
            return
        host, ev, _ = eng.status_probe
        if not ev.query():
            return
        if (int(host[0]) & 3) == 2 and getattr(self, '_carry_handled_by_caller', False):
            return                                  # GroundingEvaluator repeats the affected videos itself (evaluator._carry_tripped)
        eng.status_probe[2] = False
        if int(host[0]) & 2:
            self._ln_carry_tripped()
        if int(host[0]) & 1:
            self.numerics_status(reset=True)
            raise RuntimeError("an activation left the fp16 operand range of the f16x3 GEMM mode (|a| >= 4094) in an earlier "
                               "forward of this model: its logits were set to NaN.  Re-run with opt.model.gemm_mode = 'bf16x6' "
                               "(or 'fp32'); the flag has been reset.")

    def _position_encoding(self, T, device):
        """vid_net.pe for length T, token-major (T, E) (video_net.py:75-78,141-151)."""
        eng = self._engine
        if T not in eng.pe_cache:
            vn = self.vid_net
            pe = sinusoid_encoding(vn.max_seq_len, vn.embd_dim // 2) / vn.embd_dim ** 0.5
            if T > vn.max_seq_len:
                pe = F.interpolate(pe[None], size=T, mode='linear', align_corners=True)[0]
            eng.pe_cache = {T: pe[:, :T].t().contiguous().to(device)}
        return eng.pe_cache[T]

    def forward_window(self, vid, shallow_vid, vid_masks, text, text_masks, gate, pe_tokens=None):
        """Extension used by T-sharding (dist.py): the eval forward on a window of a longer video with an
        externally selected 0/1 clip ``gate`` (NQ, T) and the window's slice ``pe_tokens`` (T, E) of the whole
        video's position encoding.  Same return structure as ``forward(..., eval=True)``."""
        return self._drop_forward_eval(vid, shallow_vid, vid_masks, text, None, text_masks, eval=True, gate=gate,
                                       pe_tokens=pe_tokens)

    # -- one long video cut at pyramid level k (dist.hybrid_forward; dcf_hybrid_phase1 / 2 / 3) --------------------------------
    def hybrid_phase1(self, vid_w, shallow_w, mask_w, text, text_masks, gate_w, k, Tc, pe_tokens=None):
        """Levels 0..k on the narrow window (vid_w, shallow_w: (D, Tn); mask_w: (Tn,); gate_w: (NQ, Tn) the externally selected gate;
        pe_tokens: (Tn, E) the window's slice of the video's position encoding; Tc: level-k rows of the coarse window the later phases
        will use).  Returns the window's level-k features (NQ, Tn >> k, E)."""
        dev = vid_w.device
        if self._engine is None:
            self._engine = _Engine(self._config())
        self._raise_if_flagged()
        eng = self._engine
        eng.bind(self)
        lib = eng.lib
        Tn = vid_w.size(-1)
        vid_c, sh_c = vid_w.contiguous().float(), shallow_w.contiguous().float()
        mask_c = mask_w.reshape(-1).to(torch.bool).contiguous()
        nq = len(text)
        gate_c = gate_w.contiguous().float()
        assert vid_c.shape == sh_c.shape == (self.D, Tn) and mask_c.numel() == Tn and gate_c.shape == (nq, Tn)
        keep = []
        tptr, mptr, tlen = (ctypes.c_void_p * nq)(), (ctypes.c_void_p * nq)(), (ctypes.c_int32 * nq)()
        for q in range(nq):
            t = text[q][0].contiguous().float()
            m = text_masks[q].reshape(-1).to(torch.bool).contiguous()
            keep += [t, m]
            tptr[q], mptr[q], tlen[q] = t.data_ptr(), m.data_ptr(), t.size(1)
        pe = None
        if self.vid_net.use_abs_pe:
            pe = pe_tokens.contiguous().float()
            assert pe.shape == (Tn, self.E)
            _lib.check(lib.dcf_model_set_pe(eng.handle, _lib.ptr(pe), Tn), 'dcf_model_set_pe')
        featk = torch.empty(nq, Tn >> k, self.E, device=dev, dtype=torch.float32)
        _lib.check(lib.dcf_hybrid_phase1(eng.handle, k, _lib.ptr(vid_c), _lib.ptr(sh_c), _lib.ptr(mask_c), Tn, Tc, nq, tptr, mptr, tlen,
                                         _lib.ptr(gate_c), _lib.ptr(featk), _lib.current_stream()), 'dcf_hybrid_phase1')
        self._last_inputs = (vid_c, sh_c, mask_c, gate_c, keep, pe)
        self._hybrid = (k, Tn, Tc, nq)
        return featk

    def hybrid_phase2(self, featk_c, maskk_c, off_k):
        """featk_c (NQ, Tc, E), maskk_c (Tc,): the gathered level-k features / level-k validity on the coarse window; off_k: first level-k
        row of the narrow window inside it.  Returns the refined map at level k on the narrow window (NQ, Tn >> k, 32)."""
        k, Tn, Tc, nq = self._hybrid
        lib, eng = self._engine.lib, self._engine
        f = featk_c.contiguous().float()
        mk = maskk_c.reshape(-1).to(torch.bool).contiguous()
        assert f.shape == (nq, Tc, self.E) and mk.numel() == Tc
        refk = torch.empty(nq, Tn >> k, 32, device=f.device, dtype=torch.float32)
        _lib.check(lib.dcf_hybrid_phase2(eng.handle, _lib.ptr(f), _lib.ptr(mk), int(off_k), _lib.ptr(refk), _lib.current_stream()),
                   'dcf_hybrid_phase2')
        self._last_inputs = self._last_inputs + (f, mk)
        return refk

    def hybrid_phase3(self, refk_c):
        """refk_c (NQ, Tc, 32): the gathered refined level-k map on the coarse window.  Returns ((logits, offsets, masks) of levels 0..k
        on the narrow window, the same of levels k+1.. on the coarse window), each flat over its levels like ``_last_flat``."""
        k, Tn, Tc, nq = self._hybrid
        lib, eng = self._engine.lib, self._engine
        r = refk_c.contiguous().float()
        assert r.shape == (nq, Tc, 32)
        L = self.vid_net.arch[2]
        Sn = sum(Tn >> l for l in range(k + 1))
        Sc = sum(Tc >> j for j in range(1, L - k))
        dev = r.device
        ln, on, mn = (torch.empty(nq, Sn, device=dev), torch.empty(nq, Sn, 2, device=dev), torch.empty(nq, Sn, device=dev, dtype=torch.bool))
        lc, oc, mc = (torch.empty(nq, max(Sc, 1), device=dev), torch.empty(nq, max(Sc, 1), 2, device=dev),
                      torch.empty(nq, max(Sc, 1), device=dev, dtype=torch.bool))
        _lib.check(lib.dcf_hybrid_phase3(eng.handle, _lib.ptr(r), _lib.ptr(ln), _lib.ptr(on), _lib.ptr(mn), _lib.ptr(lc), _lib.ptr(oc),
                                         _lib.ptr(mc), _lib.current_stream()), 'dcf_hybrid_phase3')
        self._last_inputs = self._last_inputs + (r,)
        self._probe_numerics()
        return (ln, on, mn), (lc[:, :Sc], oc[:, :Sc], mc[:, :Sc])

    def full_position_encoding(self, T, device):
        """vid_net.pe for a video of T clips, token-major (T, E)"""
        if self._engine is None:
            self._engine = _Engine(self._config())
        return self._position_encoding(T, device)

    def forward_videos(self, videos):
        """Throughput extension (no reference counterpart: model.py:496 asserts one video per call): several videos of the
        SAME padded length in one forward (dcf_forward_eval_videos).  ``videos``: sequence of tuples
        ``(vid (1,D,T), shallow_vid (1,D,T), vid_masks (1,T), text: tuple, text_cls (NQ,D), text_masks: tuple)`` --
        the arguments of ``forward`` per video.  Returns a list with, per video, what ``forward(..., eval=True)`` returns."""
        assert len(videos) >= 1
        if not videos[0][0].is_cuda:
            raise RuntimeError('the grounding forward runs on the MI355X only: move the inputs to the GPU')
        dev = videos[0][0].device
        if self._engine is None:
            self._engine = _Engine(self._config())
        self._raise_if_flagged()
        eng = self._engine
        eng.bind(self)
        lib = eng.lib
        T = videos[0][0].size(-1)
        nv = len(videos)
        keep = []
        vptr, sptr, mptr_v, cptr = ((ctypes.c_void_p * nv)() for _ in range(4))
        nqs = (ctypes.c_int32 * nv)()
        flat_text, flat_masks = [], []
        for v, (vid, shallow, vmask, text, text_cls, tmasks) in enumerate(videos):
            assert vid.size(0) == 1 and vid.size(-1) == T, 'videos of one call share the padded length'
            vc, sc = vid[0].contiguous().float(), shallow[0].contiguous().float()
            mc = vmask.reshape(-1).to(torch.bool).contiguous()
            if not isinstance(text, (tuple, list)):
                text, tmasks = (text,), (tmasks,)
            cc = text_cls.contiguous().float()
            assert sc.shape == vc.shape == (self.D, T) and mc.numel() == T and cc.shape == (len(text), self.D)
            keep += [vc, sc, mc, cc]
            vptr[v], sptr[v], mptr_v[v], cptr[v], nqs[v] = vc.data_ptr(), sc.data_ptr(), mc.data_ptr(), cc.data_ptr(), len(text)
            flat_text += list(text)
            flat_masks += list(tmasks)
        nq = len(flat_text)
        tptr, mptr, tlen = (ctypes.c_void_p * nq)(), (ctypes.c_void_p * nq)(), (ctypes.c_int32 * nq)()
        for q in range(nq):
            t = flat_text[q][0].contiguous().float()
            m = flat_masks[q].reshape(-1).to(torch.bool).contiguous()
            assert t.dim() == 2 and m.numel() == t.size(1)
            keep += [t, m]
            tptr[q], mptr[q], tlen[q] = t.data_ptr(), m.data_ptr(), t.size(1)
        pe = None
        if self.vid_net.use_abs_pe:
            Tp = T // self.vid_net.stride            # the pyramid (and its position encoding) starts behind the strided convolutions
            pe = self._position_encoding(Tp, dev)
            _lib.check(lib.dcf_model_set_pe(eng.handle, _lib.ptr(pe), Tp), 'dcf_model_set_pe')
        S = lib.dcf_points_per_query(eng.handle, T)
        okey = (nq, S, dev)
        if self.reuse_output_buffers and okey in self._out_cache:
            logits, offsets, masks = self._out_cache[okey]
        else:
            logits = torch.empty(nq, S, device=dev, dtype=torch.float32)
            offsets = torch.empty(nq, S, 2, device=dev, dtype=torch.float32)
            masks = torch.empty(nq, S, device=dev, dtype=torch.bool)
            if self.reuse_output_buffers:
                self._out_cache = {okey: (logits, offsets, masks)}
        _lib.check(lib.dcf_forward_eval_videos(eng.handle, nv, vptr, sptr, mptr_v, T, nqs, tptr, mptr, tlen, cptr, _lib.ptr(logits),
                                               _lib.ptr(offsets), _lib.ptr(masks), _lib.current_stream()), 'dcf_forward_eval_videos')
        self._last_inputs = (keep, pe)
        self._last_flat = (logits, offsets, masks)
        self._probe_numerics()
        L = self.vid_net.arch[2]
        sizes = [(T // self.vid_net.stride) >> l for l in range(L)]
        out, q = [], 0
        for v in range(nv):
            n = nqs[v]
            out.append(([tuple(x.unsqueeze(0) for x in logits[i].split(sizes)) for i in range(q, q + n)],
                        [tuple(x.unsqueeze(0) for x in offsets[i].split(sizes)) for i in range(q, q + n)],
                        [tuple(x.unsqueeze(0) for x in masks[i].split(sizes)) for i in range(q, q + n)]))
            q += n
        return out

    def _drop_forward_eval(self, vid, shallow_vid, vid_masks, text, text_cls, text_masks, text_size=None, mv_data=None,
                           eval=False, gate=None, pe_tokens=None):
        assert mv_data is None and eval
        assert vid.size(0) == 1, vid.size()                                  # model.py:496
        if not vid.is_cuda:
            raise RuntimeError('the grounding forward runs on the MI355X only: move the inputs to the GPU')
        dev = vid.device
        if self._engine is None:
            self._engine = _Engine(self._config())
        self._raise_if_flagged()
        eng = self._engine
        eng.bind(self)
        lib = eng.lib
        T = vid.size(-1)
        vid_c = vid[0].contiguous().float()
        sh_c = shallow_vid[0].contiguous().float()
        mask_c = vid_masks.reshape(-1).to(torch.bool).contiguous()
        assert mask_c.numel() == T and sh_c.shape == vid_c.shape == (self.D, T)
        if not isinstance(text, (tuple, list)):
            text, text_masks = (text,), (text_masks,)
        nq = len(text)
        if gate is None:
            cls_c = text_cls.contiguous().float()
            assert cls_c.shape == (nq, self.D), (cls_c.shape, nq, self.D)
        else:
            cls_c = gate.contiguous().float()
            assert cls_c.shape == (nq, T), (cls_c.shape, nq, T)
        keep = []
        tptr = (ctypes.c_void_p * nq)()
        mptr = (ctypes.c_void_p * nq)()
        tlen = (ctypes.c_int32 * nq)()
        for q in range(nq):
            t = text[q][0].contiguous().float()                              # (TE, Lk)
            m = text_masks[q].reshape(-1).to(torch.bool).contiguous()
            assert t.dim() == 2 and m.numel() == t.size(1)
            keep += [t, m]
            tptr[q], mptr[q], tlen[q] = t.data_ptr(), m.data_ptr(), t.size(1)
        if self.vid_net.use_abs_pe:
            Tp = T // self.vid_net.stride
            pe = self._position_encoding(Tp, dev) if pe_tokens is None else pe_tokens.contiguous().float()
            assert pe.shape == (Tp, self.E)
            _lib.check(lib.dcf_model_set_pe(eng.handle, _lib.ptr(pe), Tp), 'dcf_model_set_pe')
        else:
            pe = None
        S = lib.dcf_points_per_query(eng.handle, T)
        okey = (nq, S, dev)
        if self.reuse_output_buffers and okey in self._out_cache:
            logits, offsets, masks = self._out_cache[okey]
        else:
            logits = torch.empty(nq, S, device=dev, dtype=torch.float32)
            offsets = torch.empty(nq, S, 2, device=dev, dtype=torch.float32)
            masks = torch.empty(nq, S, device=dev, dtype=torch.bool)
            if self.reuse_output_buffers:
                self._out_cache = {okey: (logits, offsets, masks)}
        fn = lib.dcf_forward_eval if gate is None else lib.dcf_forward_eval_gated
        _lib.check(fn(eng.handle, _lib.ptr(vid_c), _lib.ptr(sh_c), _lib.ptr(mask_c), T, nq, tptr, mptr, tlen,
                      _lib.ptr(cls_c), _lib.ptr(logits), _lib.ptr(offsets), _lib.ptr(masks), _lib.current_stream()),
                   'dcf_forward_eval')
        # keep the borrowed inputs alive until the stream has consumed them
        self._last_inputs = (vid_c, sh_c, mask_c, cls_c, keep, pe)
        self._last_flat = (logits, offsets, masks)
        self._probe_numerics()
        L = self.vid_net.arch[2]
        sizes = [(T // self.vid_net.stride) >> l for l in range(L)]
        lg = [tuple(x.unsqueeze(0) for x in logits[q].split(sizes)) for q in range(nq)]
        of = [tuple(x.unsqueeze(0) for x in offsets[q].split(sizes)) for q in range(nq)]
        mk = [tuple(x.unsqueeze(0) for x in masks[q].split(sizes)) for q in range(nq)]
        return lg, of, mk


class PtTransformer(PtTransformerEarlyFusionIterative):
    """Drop-in for the late-fusion model libs/modeling/model.py:30-161 (exported by libs/modeling/__init__.py:2): the gated
    [vid ; shallow] features feed ``vid_net`` directly (``vid_net.embd_fc`` is (E, 2D, 1)), the fusion stack is applied to
    every pyramid level and ``cls_head`` / ``reg_head`` predict from the fused pyramid.  Same forward signature."""

    MODEL_KIND = 1

    def __init__(self, opt):
        nn.Module.__init__(self)
        mo = copy.deepcopy(opt['model'] if isinstance(opt, dict) else opt.model)
        self.opt = opt
        vn, tn, fu = dict(mo['vid_net']), dict(mo['text_net']), dict(mo['fusion'])
        self.sn, self.sratio = int(mo['sn']), float(mo['sratio'])
        self.msf, self.norm = bool(mo['msf']), bool(mo['norm'])
        self.scat, self.sfonly = bool(mo.get('scat', False)), False     # model.py:30-161 never reads opt.model.sfonly
        D, E = int(vn['in_dim']), int(vn['embd_dim'])
        self.D, self.E = D, E
        self.text_net = make_text_net(tn)
        vn.pop('name', None)
        vn['in_dim'] = (2 * D if self.msf else D) + int(self.scat)      # model.py:43-48
        self.vid_net = VideoTransformer(**vn)
        fu.pop('name', None)
        self.fusion = XAttNFusion(**fu)
        ch, rh = dict(mo['cls_head']), dict(mo['reg_head'])
        n_levels = self.vid_net.arch[2]
        self.cls_head = ConvHead(ch['embd_dim'], 'cls_head', 1, ch.get('n_layers', 2), None, ch.get('prior_prob', 0.0))
        self.reg_head = ConvHead(rh['embd_dim'], 'reg_head', 2, rh.get('n_layers', 2), rh.get('num_fpn_levels', n_levels))
        self.second_fusion = False
        self.head_layers = ch.get('n_layers', 2)
        self.max_batch = int(mo.get('max_batch', 0) or 0)
        self.gemm_mode = GEMM_MODES[mo.get('gemm_mode', 'f16x3')]
        self.attn_mode = ATTN_MODES[mo.get('attn_mode', 'f16x3')]
        self._engine = None
        self.reuse_output_buffers = False
        self.ln_carry = True                        # LayerNorms carried between kernels as row statistics where the kernels allow (set_ln_carry)
        self._out_cache = {}
        self.graph_mode = 'auto'


class PtTransformerEarlyFusion(PtTransformerEarlyFusionIterative):
    """Drop-in for libs/modeling/model.py:163-373: early fusion like the iterative model (vid_map, XAttNFusion on the clip
    sequence, vid_net, optionally the fusion stack again on every pyramid level) but no refinement stage: ``cls_head`` and
    ``reg_head`` predict from the E-wide pyramid (model.py:204-209).  Same forward signature (eval, and the training-mode forward
    values at dropout 0: ``_drop_forward``)."""

    MODEL_KIND = 2

    def __init__(self, opt, second_fusion=True):
        nn.Module.__init__(self)
        mo = copy.deepcopy(opt['model'] if isinstance(opt, dict) else opt.model)
        self.opt = opt
        vn, tn, fu = dict(mo['vid_net']), dict(mo['text_net']), dict(mo['fusion'])
        self.sn, self.sratio = int(mo['sn']), float(mo['sratio'])
        self.msf, self.norm = bool(mo['msf']), bool(mo['norm'])
        self.scat, self.sfonly = bool(mo.get('scat', False)), False     # model.py:163-373 never reads opt.model.sfonly
        D, E = int(vn['in_dim']), int(vn['embd_dim'])
        self.D, self.E = D, E
        self.text_net = make_text_net(tn)
        self.vid_map = MaskedConv1D((2 * D if self.msf else D) + int(self.scat), E, 1)      # model.py:175-181
        vn.pop('name', None)
        vn['in_dim'] = E
        self.vid_net = VideoTransformer(**vn)
        fu.pop('name', None)
        self.fusion = XAttNFusion(**fu)
        ch, rh = dict(mo['cls_head']), dict(mo['reg_head'])
        n_levels = self.vid_net.arch[2]
        self.cls_head = ConvHead(ch['embd_dim'], 'cls_head', 1, ch.get('n_layers', 2), None, ch.get('prior_prob', 0.0))
        self.reg_head = ConvHead(rh['embd_dim'], 'reg_head', 2, rh.get('n_layers', 2), rh.get('num_fpn_levels', n_levels))
        self.second_fusion = second_fusion
        self.head_layers = ch.get('n_layers', 2)
        self.max_batch = int(mo.get('max_batch', 0) or 0)
        self.gemm_mode = GEMM_MODES[mo.get('gemm_mode', 'f16x3')]
        self.attn_mode = ATTN_MODES[mo.get('attn_mode', 'f16x3')]
        self._engine = None
        self.reuse_output_buffers = False
        self.ln_carry = True                        # LayerNorms carried between kernels as row statistics where the kernels allow (set_ln_carry)
        self._out_cache = {}
        self.graph_mode = 'auto'


def create_model(opt):
    """libs/worker_v2.py:182-211: only ``opt.model.name == 'iter'`` exists in the reference."""
    name = opt['model']['name'] if isinstance(opt, dict) else opt.model.name
    if name != 'iter':
        raise NotImplementedError(f"unknown model name {name!r}: the reference only builds 'iter'")
    return PtTransformerEarlyFusionIterative(opt, second_fusion=False)


class PtGenerator(nn.Module):
    """Candidate point generator, model.py:668-743 (same ctor, buffers and forward)."""

    def __init__(self, max_seq_len, num_fpn_levels, regression_range=4, sigma=1, use_offset=False):
        super().__init__()
        self.num_fpn_levels = num_fpn_levels
        assert max_seq_len % 2 ** (num_fpn_levels - 1) == 0
        self.max_seq_len = max_seq_len
        assert 0 < sigma <= 1
        rng = [(0, regression_range)]
        rr = regression_range
        for l in range(1, num_fpn_levels):
            assert rr <= max_seq_len
            lo, hi = rr * sigma, rr * 2
            if l == num_fpn_levels - 1:
                hi = max(hi, max_seq_len + 1)
            rng.append((lo, hi))
            rr = hi
        self.regression_range = tuple(rng)
        self.use_offset = use_offset
        tics = torch.arange(0, max_seq_len, 1.0)
        for l in range(num_fpn_levels):
            stride = 2 ** l
            pts = tics[::stride][:, None].clone()
            if use_offset:
                pts += 0.5 * stride
            r = torch.as_tensor(rng[l], dtype=torch.float32)[None].repeat(len(pts), 1)
            s = torch.full((len(pts), 1), float(stride))
            self.register_buffer(f'points_{l}', torch.cat((pts, r, s), 1), persistent=False)

    @property
    def buffer_points(self):
        return [getattr(self, f'points_{l}') for l in range(self.num_fpn_levels)]

    def forward(self, fpn_n_points):
        assert len(fpn_n_points) == self.num_fpn_levels
        out = tuple()
        for n, pts in zip(fpn_n_points, self.buffer_points):
            assert n <= len(pts), f'number of requested points {n} cannot exceed max number of buffered points {len(pts)}'
            out += (pts[:n],)
        return out
