"""CPU ORACLE (test infrastructure, NOT product code) for the DeCafNet grounding hot path.

This file is a from-scratch, functional fp32 PyTorch-CPU restatement of the reference
algorithm.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` may import it; the product path (``cvpr2025-decafnet_amd``) never does.

Parity status: PINNED.  Every function below is checked in ``tests/test_oracle_golden.py``
against fixtures under ``tests/golden/`` that were produced by importing the real
reference from ``/root/reference`` (generator: ``tests/golden/make_golden.py``).

All tensors use the reference layout: features ``(bs, C, T)`` channel-major, masks
``(bs, 1, T)`` bool.  Weights are looked up in a flat ``state_dict`` with the
reference's parameter names.

Reference citations are relative to /root/reference/.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ----------------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------------
def channel_layer_norm(x: Tensor, weight=None, bias=None, eps: float = 1e-5) -> Tensor:
    """LayerNorm over the channel axis of (bs, C, T).  libs/modeling/blocks.py:125-131.

    Biased variance, eps inside the sqrt, affine params shaped (C, 1).
    """
    x = x - x.mean(dim=1, keepdim=True)
    var = (x * x).mean(dim=1, keepdim=True)
    x = x / torch.sqrt(var + eps)
    if weight is not None:
        x = x * weight + bias
    return x


def masked_conv1d(x: Tensor, mask: Tensor, weight: Tensor, bias=None, stride: int = 1,
                  padding: int = 0, groups: int = 1) -> Tuple[Tensor, Tensor]:
    """conv1d(x * mask); output NOT re-masked; stride>1 downsamples the mask by
    nearest (== mask[..., ::stride]).  libs/modeling/blocks.py:87-106."""
    assert x.size(-1) % stride == 0
    mf = mask.to(x.dtype)
    y = F.conv1d(x * mf, weight, bias, stride=stride, padding=padding, groups=groups)
    if stride > 1:
        mf = F.interpolate(mf, size=y.size(-1), mode='nearest')
        mask = mf.bool()
    return y, mask


def masked_max_pool1d(x: Tensor, mask: Tensor, kernel_size: int = 3, stride: int = 2):
    """libs/modeling/blocks.py:31-47: padded slots are filled with the per-channel global
    minimum, then maxpool(k, s, pad=(k-1)//2) on x and on the mask, re-mask."""
    x_min = x.amin(dim=-1, keepdim=True)
    mf = mask.to(x.dtype)
    x = x * mf + (~mask).to(x.dtype) * x_min
    pad = (kernel_size - 1) // 2
    x = F.max_pool1d(x, kernel_size, stride, pad)
    mf = F.max_pool1d(mf, kernel_size, stride, pad)
    return x * mf, mf.to(mask.dtype)


def sinusoid_encoding(seq_len: int, n_freqs: int) -> Tensor:
    """libs/modeling/blocks.py:134-142."""
    tics = torch.arange(seq_len, dtype=torch.float)
    freqs = 10000 ** torch.linspace(0, 1, n_freqs + 1)[:n_freqs]
    ang = tics[None, :] / freqs[:, None]
    return torch.cat((torch.sin(ang), torch.cos(ang)))


def position_encoding(max_seq_len: int, embd_dim: int) -> Tensor:
    """video_net.py:75-78 / text_net.py:121-124: PE / sqrt(E), shape (E, max_seq_len)."""
    pe = sinusoid_encoding(max_seq_len, embd_dim // 2)
    return pe / embd_dim ** 0.5


def resample_pe(pe: Tensor, t: int, max_seq_len: int) -> Tensor:
    """Eval-time PE: linear resample (align_corners) when t > max_seq_len, else crop.
    video_net.py:141-151."""
    if t > max_seq_len:
        pe = F.interpolate(pe[None], size=t, mode='linear', align_corners=True)[0]
    return pe[..., :t]


def _project(x: Tensor, sd: SD, prefix: str) -> Tensor:
    return F.conv1d(x, sd[prefix + '.weight'], sd.get(prefix + '.bias'))


def mha_global(sd: SD, p: str, q_in: Tensor, kv_in: Tensor, kv_mask: Tensor, n_heads: int) -> Tensor:
    """MaskedMHA global cross-attention (k = v = kv), blocks.py:339-356,374-393."""
    return _mha_global_qkv(sd, p, q_in, kv_in, kv_in, kv_mask, n_heads)


def banded_attention(q: Tensor, k: Tensor, v: Tensor, mask: Tensor, window: int) -> Tensor:
    """Sliding-window attention core written as an explicit band (the reference realises
    the same arithmetic with overlapping chunks, blocks.py:204-325):

        out_t = sum_{|j-t| <= w//2, 0 <= j < T} softmax_j(q_t.k_j + pen_j) v_j
        pen_j = -1e4 if key j is padded (blocks.py:279), out-of-range keys = -inf (:260-261),
        rows of padded queries are forced to 0 (:293).

    q, k, v: (n, T, d) already scaled; mask: (bs, T) bool with n = bs * heads.
    """
    n, t, d = q.shape
    s = window // 2
    bs = mask.size(0)
    h = n // bs
    kp = F.pad(k, (0, 0, s, s))
    vp = F.pad(v, (0, 0, s, s))
    kw = kp.unfold(1, window, 1)            # (n, T, d, w)
    vw = vp.unfold(1, window, 1)            # (n, T, d, w)
    att = torch.einsum('ntd,ntdw->ntw', q, kw)
    pos = torch.arange(t)[:, None] + torch.arange(-s, s + 1)[None, :]      # (T, w)
    in_range = (pos >= 0) & (pos < t)
    key_valid = F.pad(mask, (s, s)).unfold(1, window, 1)                     # (bs, T, w)
    pen = torch.zeros(bs, t, window, dtype=q.dtype)
    pen = pen.masked_fill(~key_valid, -1e4)
    pen = pen.masked_fill(~in_range[None], float('-inf'))
    att = att.view(bs, h, t, window) + pen[:, None]
    att = F.softmax(att, dim=-1)
    att = att.masked_fill(~mask[:, None, :, None], 0.0)
    out = torch.einsum('ntw,ntdw->ntd', att.view(n, t, window), vw)
    return out


def mha_local(sd: SD, p: str, q_in: Tensor, k_in: Tensor, v_in: Tensor, mask: Tensor,
              n_heads: int, window: int) -> Tensor:
    """MaskedMHA local branch, blocks.py:357-373 (+ proj :391-392).  mask: (bs,1,T)."""
    q = _project(q_in, sd, p + '.query')
    k = _project(k_in, sd, p + '.key')
    v = _project(v_in, sd, p + '.value')
    bs, c, t = q.shape
    d = c // n_heads
    scale = 1.0 / math.sqrt(math.sqrt(d))
    assert t % max(window // 2, 1) == 0, 'blocks.py:216 needs T % (w//2) == 0'

    def split(z):
        return z.view(bs, n_heads, d, t).flatten(0, 1).transpose(1, 2)

    out = banded_attention(split(q) * scale, split(k) * scale, split(v), mask[:, 0], window)
    out = out.view(bs, n_heads, t, d).transpose(2, 3).reshape(bs, c, t)
    return _project(out, sd, p + '.proj')


def ffn(sd: SD, p: str, x: Tensor) -> Tensor:
    """blocks.py:535-538: 1x1 E->4E, exact (erf) GELU, 1x1 4E->E."""
    return _project(F.gelu(_project(x, sd, p + '.fc')), sd, p + '.proj')


def _ln(sd: SD, p: str, x: Tensor) -> Tensor:
    return channel_layer_norm(x, sd[p + '.weight'], sd[p + '.bias'])


def transformer_encoder(sd: SD, p: str, x: Tensor, mask: Tensor, stride: int, n_heads: int,
                        window: int) -> Tuple[Tensor, Tensor]:
    """TransformerEncoder.forward blocks.py:578-591 + ConvAttNLayer.forward :462-473.
    stride 0 = no depthwise convs (text encoder); window 0 = global attention."""
    mf = mask.to(x.dtype)
    x = x * mf
    skip = masked_max_pool1d(x, mask, 3, stride)[0] if stride > 1 else x
    xn = _ln(sd, p + '.ln_attn', x)
    if stride > 0:
        e = x.size(1)
        k, _ = masked_conv1d(xn, mask, sd[p + '.attn.k_conv.conv.weight'], None, stride, 1, e)
        v, _ = masked_conv1d(xn, mask, sd[p + '.attn.v_conv.conv.weight'], None, stride, 1, e)
        q, mask = masked_conv1d(xn, mask, sd[p + '.attn.q_conv.conv.weight'], None, stride, 1, e)
        q = _ln(sd, p + '.attn.q_norm', q)
        k = _ln(sd, p + '.attn.k_norm', k)
        v = _ln(sd, p + '.attn.v_norm', v)
    else:
        q = k = v = xn
    if window > 0:
        h = mha_local(sd, p + '.attn.attn', q, k, v, mask, n_heads, window)
    else:
        # self-attention, global: blocks.py:339-343 (k = q, v = k inputs)
        h = _mha_global_qkv(sd, p + '.attn.attn', q, k, v, mask, n_heads)
    mf = mask.to(x.dtype)
    x = skip * mf + sd[p + '.drop_path_attn.scale'] * h
    h = ffn(sd, p + '.ffn', _ln(sd, p + '.ln_ffn', x)) * mf
    x = x + sd[p + '.drop_path_ffn.scale'] * h
    return x, mask


def _mha_global_qkv(sd: SD, p: str, q_in, k_in, v_in, kv_mask, n_heads):
    """MaskedMHA global branch, blocks.py:348-356,374-393.  q (bs,Cq,T1), k/v (bs,Ckv,T2),
    kv_mask (bs,1,T2).  Both q and k are scaled by d_head**-0.25 before the product."""
    q = _project(q_in, sd, p + '.query')
    k = _project(k_in, sd, p + '.key')
    v = _project(v_in, sd, p + '.value')
    bs, c, _ = q.shape
    d = c // n_heads
    scale = 1.0 / math.sqrt(math.sqrt(d))
    q = q.view(bs, n_heads, d, -1).transpose(2, 3)
    k = k.view(bs, n_heads, d, -1)
    v = v.view(bs, n_heads, d, -1).transpose(2, 3)
    att = (q * scale) @ (k * scale)
    att = att.masked_fill(~kv_mask[:, :, None, :], float('-inf'))
    att = F.softmax(att, dim=-1)
    out = (att @ v).transpose(2, 3).reshape(bs, c, -1)
    return _project(out, sd, p + '.proj')


def transformer_decoder(sd: SD, p: str, q: Tensor, q_mask: Tensor, kv: Tensor, kv_mask: Tensor,
                        n_heads: int, adaln: bool = True) -> Tuple[Tensor, Tensor]:
    """TransformerDecoder.forward blocks.py:632-650 + ConvXAttNLayer.forward :513-520."""
    qf = q_mask.to(q.dtype)
    q = q * qf
    qn = _ln(sd, p + '.ln_xattn_q', q)
    kvn = _ln(sd, p + '.ln_xattn_kv', kv)
    e = q.size(1)
    qc, _ = masked_conv1d(qn, q_mask, sd[p + '.xattn.q_conv.conv.weight'], None, 1, 1, e)
    qc = _ln(sd, p + '.xattn.q_norm', qc)
    h = mha_global(sd, p + '.xattn.xattn', qc, kvn, kv_mask, n_heads)        # (bs, 2E, T)
    q = q * qf
    if adaln:
        q = channel_layer_norm(q)
    scale, shift = h.chunk(2, dim=1)
    q = q * scale + shift
    h = ffn(sd, p + '.ffn', _ln(sd, p + '.ln_ffn', q)) * qf
    q = q + sd[p + '.drop_path_ffn.scale'] * h
    return q, q_mask


# ----------------------------------------------------------------------------------
# sub-networks
# ----------------------------------------------------------------------------------
def text_transformer(sd: SD, cfg, tokens: Tensor, token_mask: Tensor, p: str = 'text_net'):
    """TextTransformer.forward text_net.py:158-188 (eval).  tokens (bs,C_t,Lq)."""
    bs, _, t = tokens.shape
    mask = token_mask if token_mask.ndim == 3 else token_mask.unsqueeze(1)
    x, _ = masked_conv1d(tokens, mask, sd[p + '.embd_fc.conv.weight'], sd[p + '.embd_fc.conv.bias'])
    if cfg['use_abs_pe']:
        pe = position_encoding(cfg['max_seq_len'], cfg['embd_dim'])
        x = x + resample_pe(pe, t, cfg['max_seq_len']) * mask.to(x.dtype)
    if cfg.get('use_bkgd_token', True):
        x = torch.cat((sd[p + '.bkgd_token'].repeat(bs, 1, 1), x), dim=-1)
        mask = torch.cat((mask[..., :1], mask), dim=-1)
    for i in range(cfg.get('n_layers', 5)):
        x, _ = transformer_encoder(sd, f'{p}.transformer.{i}', x, mask, 0, cfg['n_heads'], 0)
    return x, mask


def xattn_fusion(sd: SD, cfg, vid: Tensor, vid_mask: Tensor, text: Tensor, text_mask: Tensor,
                 p: str = 'fusion'):
    """XAttNFusion._forward fusion.py:56-66."""
    for i in range(cfg.get('n_layers', 2)):
        vid, vid_mask = transformer_decoder(sd, f'{p}.layers.{i}', vid, vid_mask, text, text_mask,
                                            cfg.get('n_heads', 4), cfg.get('xattn_mode', 'adaln') == 'adaln')
    return _ln(sd, p + '.ln_out', vid), vid_mask


def xattn_fusion_pyramid(sd: SD, cfg, fpn, fpn_masks, text: Tensor, text_mask: Tensor, p: str = 'fusion'):
    """XAttNFusion.forward on an FPN tuple, fusion.py:68-78: the same decoder stack applied to every level."""
    out, out_masks = tuple(), tuple()
    for x, m in zip(fpn, fpn_masks):
        y, ym = xattn_fusion(sd, cfg, x, m, text, text_mask, p)
        out += (y,)
        out_masks += (ym,)
    return out, out_masks


def video_transformer(sd: SD, cfg, x: Tensor, mask: Tensor, p: str = 'vid_net', pe_override=None):
    """VideoTransformer.forward video_net.py:123-164 (eval).  ``cfg.stride`` = s > 1: the first log2(s) embedding convolutions are
    k5 / stride 2 / padding 2 (video_net.py:62-73), each halving the sequence and its mask; ``cfg.pool_only``: a branch layer is one
    depthwise k3 MaskedConv1D (stride 1 at level 0, 2 above) instead of a TransformerEncoder (video_net.py:98-111).
    ``pe_override`` (E, T): a window's slice of a longer video's position encoding (T-sharding tests)."""
    if mask.ndim == 2:
        mask = mask.unsqueeze(1)
    stride = int(cfg.get('stride', 1))
    assert stride >= 1 and stride & (stride - 1) == 0
    n_convs, n_stem, n_branch = cfg['arch']
    x, _ = masked_conv1d(x, mask, sd[p + '.embd_fc.conv.weight'], sd[p + '.embd_fc.conv.bias'])
    for i in range(n_convs):
        if stride > 1:
            x, mask = masked_conv1d(x, mask, sd[f'{p}.embd_convs.{i}.conv.weight'], None, 2, 2)
        else:
            x, mask = masked_conv1d(x, mask, sd[f'{p}.embd_convs.{i}.conv.weight'], None, 1, 1)
        x = F.relu(_ln(sd, f'{p}.embd_norms.{i}', x))
        stride = max(stride // 2, 1)
    assert stride == 1, 'vid_net.arch[0] < log2(vid_net.stride) (video_net.py:53)'
    t = x.size(-1)
    if cfg['use_abs_pe']:
        if pe_override is None:
            pe = position_encoding(cfg['max_seq_len'], cfg['embd_dim'])
            x = x + resample_pe(pe, t, cfg['max_seq_len']) * mask.to(x.dtype)
        else:
            x = x + pe_override * mask.to(x.dtype)
    for i in range(n_stem):
        x, mask = transformer_encoder(sd, f'{p}.stem.{i}', x, mask, 1, cfg['n_heads'], cfg['mha_win_size'])
    fpn, fpn_masks = tuple(), tuple()
    for i in range(n_branch):
        if cfg.get('pool_only', False):
            x, mask = masked_conv1d(x, mask, sd[f'{p}.branch.{i}.conv.weight'], None, 2 if i > 0 else 1, 1, x.size(1))
        else:
            x, mask = transformer_encoder(sd, f'{p}.branch.{i}', x, mask, 2 if i > 0 else 1,
                                          cfg['n_heads'], cfg['mha_win_size'])
        fpn += (x,)
        fpn_masks += (mask,)
    return fpn, fpn_masks


def conv_head(sd: SD, p: str, out_name: str, fpn: Sequence[Tensor], fpn_masks: Sequence[Tensor],
              n_layers: int):
    """Shared trunk of ClsHead/RegHead: n x (k3 conv no-bias, LN, ReLU) then a k3 conv.
    head.py:53-64, :95-108.  Returns the raw conv outputs per level."""
    outs = []
    for x, mask in zip(fpn, fpn_masks):
        for i in range(n_layers):
            x, _ = masked_conv1d(x, mask, sd[f'{p}.convs.{i}.conv.weight'], None, 1, 1)
            x = F.relu(_ln(sd, f'{p}.norms.{i}', x))
        y, _ = masked_conv1d(x, mask, sd[f'{p}.{out_name}.conv.weight'], sd[f'{p}.{out_name}.conv.bias'], 1, 1)
        outs.append(y)
    return outs


def cls_head(sd: SD, p: str, fpn, fpn_masks, n_layers: int = 2):
    outs = conv_head(sd, p, 'cls_head', fpn, fpn_masks, n_layers)
    return tuple(o.squeeze(1) for o in outs), tuple(m.squeeze(1) for m in fpn_masks)


def reg_head(sd: SD, p: str, fpn, fpn_masks, n_layers: int = 2):
    outs = conv_head(sd, p, 'reg_head', fpn, fpn_masks, n_layers)
    offs = tuple(F.relu(o * sd[f'{p}.scales.{i}.scale']).transpose(1, 2) for i, o in enumerate(outs))
    return offs, tuple(m.squeeze(1) for m in fpn_masks)


def tcn_refine(sd: SD, p: str, x: Tensor, mask: Tensor, n_layers: int) -> Tensor:
    """TCN.forward tcn.py:66-84 with DilatedResidualLayer.forward tcn.py:21-38 (eval:
    dropout is identity).  x (bs, L, T0); mask (bs, 1, T0) bool."""
    mf = mask.to(x.dtype)
    out = F.conv1d(x, sd[p + '.conv_1x1.weight'], sd[p + '.conv_1x1.bias'])
    for i in range(n_layers):
        dil = 2 ** i
        q = f'{p}.layers.{i}'
        h = F.relu(F.conv1d(out, sd[q + '.conv_dilated.weight'], sd[q + '.conv_dilated.bias'],
                            padding=dil, dilation=dil))
        h = F.conv1d(h, sd[q + '.conv_1x1.weight'], sd[q + '.conv_1x1.bias'])
        out = (out + h) * mf[:, 0:1, :]
        out = F.layer_norm(out.permute(0, 2, 1), (out.size(1),), sd[q + '.norm.weight'],
                           sd[q + '.norm.bias'], 1e-5).permute(0, 2, 1)
    out = F.conv1d(out, sd[p + '.conv_out.weight'], sd[p + '.conv_out.bias'])
    return out * mf[:, 0:1, :]


# ----------------------------------------------------------------------------------
# sidekick gate
# ----------------------------------------------------------------------------------
def sidekick_scores(shallow_vid: Tensor, text_cls: Tensor, norm: bool) -> Tensor:
    """model.py:500-505: cosine (eps 1e-4 added to each norm) or raw dot.  -> (NQ, T)."""
    if norm:
        v = shallow_vid / (shallow_vid.norm(dim=1, keepdim=True) + 1e-4)
        t = text_cls / (text_cls.norm(dim=1, keepdim=True) + 1e-4)
        return torch.einsum('bht,bh->bt', v, t)
    return torch.einsum('bht,bh->bt', shallow_vid, text_cls)


def topk_block_gate(correl_row: Tensor, vid_len: int, sn: int, sratio: float) -> Tensor:
    """model.py:531-541 for one query.  correl_row (T,), returns a float 0/1 gate of
    length ``vid_len``.  Ties in the ascending argsort are broken by ``torch.argsort``
    (unspecified; fixtures are tie-free)."""
    pooled = F.avg_pool1d(correl_row[None, None, :vid_len], kernel_size=sn, stride=sn, ceil_mode=True)[0, 0]
    ranked = pooled.argsort()
    k = int(sratio * pooled.shape[0])
    top = ranked[-k:]                       # k == 0  ->  ranked[-0:] == everything
    weight = torch.zeros_like(pooled)
    weight[top] = 1
    return F.interpolate(weight[None, None, :], size=vid_len, mode='nearest')[0, 0]


def gate_reference_formula(pooled: Tensor, vid_len: int, sratio: float) -> Tensor:
    """The same gate in closed form (used to document the index arithmetic the HIP kernel
    must reproduce): stable rank-by-count selection + float32 nearest index
    ``min(floor(t * float32(n / len)), n - 1)`` (ATen upsample_nearest1d)."""
    n = pooled.numel()
    k = int(sratio * n)
    if k == 0:
        sel = torch.ones(n, dtype=torch.bool)
    else:
        # rank = number of blocks strictly smaller, ties resolved by index (stable)
        lt = (pooled[None, :] < pooled[:, None]) | ((pooled[None, :] == pooled[:, None]) &
                                                    (torch.arange(n)[None, :] < torch.arange(n)[:, None]))
        rank = lt.sum(1)
        sel = rank >= n - k
    t = torch.arange(vid_len, dtype=torch.float32)
    if vid_len == n:
        idx = torch.arange(vid_len)
    elif vid_len == 2 * n:
        idx = torch.arange(vid_len) >> 1
    else:
        scale = torch.tensor(n, dtype=torch.float32) / torch.tensor(vid_len, dtype=torch.float32)
        idx = torch.clamp(torch.floor(t * scale).long(), max=n - 1)
    return sel[idx].to(torch.float32)


# ----------------------------------------------------------------------------------
# full model: PtTransformerEarlyFusionIterative eval forward
# ----------------------------------------------------------------------------------
def fuse_and_predict(sd: SD, cfg, fpn, fpn_masks, text=None, text_masks=None, second_fusion: bool = False):
    """model.py:442-471; ``second_fusion`` re-applies the (shared) fusion stack to every pyramid level (:443-444)."""
    n_levels = cfg['vid_net']['arch'][2]
    if second_fusion:
        fpn, fpn_masks = xattn_fusion_pyramid(sd, cfg['fusion'], fpn, fpn_masks, text, text_masks)
    logits1, _ = cls_head(sd, 'cls_head', fpn, fpn_masks, cfg['cls_head'].get('n_layers', 2))
    ref_len = logits1[0].shape[1]
    expand = [logits1[0]]
    for l in logits1[1:]:
        up = F.interpolate(l.unsqueeze(1), size=ref_len, mode='nearest')[:, 0]
        expand.append(up * fpn_masks[0][:, 0])
    expand = torch.stack(expand, dim=1)
    expand = tcn_refine(sd, 'refine', expand, fpn_masks[0], n_levels)
    new_fpn = []
    for i, f in enumerate(fpn):
        if i != 0:
            expand = masked_max_pool1d(expand, fpn_masks[i - 1])[0]
        new_fpn.append(torch.cat([f, expand], dim=1))
    logits2, _ = cls_head(sd, 'cls_head2', new_fpn, fpn_masks, cfg['cls_head'].get('n_layers', 2))
    offsets, out_masks = reg_head(sd, 'reg_head', new_fpn, fpn_masks, cfg['reg_head'].get('n_layers', 2))
    return logits1, logits2, offsets, out_masks


def text_identity(sd: SD, cfg, tokens: Tensor, token_mask: Tensor, p: str = 'text_net'):
    """TextIdentity.forward (text_net.py:62-89) with AttNPool1D (blocks.py:396-411) and masked_avg_pool1d (blocks.py:10-17)."""
    x, mask = tokens, token_mask
    if mask.dim() == 2:
        mask = mask.unsqueeze(1)
    t = x.size(-1)
    if f'{p}.embd_fc.conv.weight' in sd:
        x, _ = masked_conv1d(x, mask, sd[f'{p}.embd_fc.conv.weight'], sd.get(f'{p}.embd_fc.conv.bias'))
    if cfg.get('use_abs_pe', False):
        e = x.size(1)
        pe = position_encoding(cfg['max_seq_len'], e)
        if t > cfg['max_seq_len']:
            pe = resample_pe(pe, t, cfg['max_seq_len'])
        x = x + pe[..., :t] * mask.to(x.dtype)
    if cfg.get('use_bkgd_token', True):
        x_mean = torch.sum(x * mask.to(x.dtype), dim=-1, keepdim=True) / torch.sum(mask, dim=-1, keepdim=True)
        h = torch.cat((x_mean, x), dim=-1)
        mask = torch.cat((mask[..., :1], mask), dim=-1)
        pool = mha_global(sd, f'{p}.attn_pool.attn', h, h, mask, cfg.get('n_heads', 4))[..., :1]
        x = torch.cat((pool, x), dim=-1)
    return x, mask


def encode_text(sd: SD, cfg, tokens: Tensor, token_masks: Tensor):
    """model.py:434-436 (make_text_net, text_net.py:191-193: 'transformer' or 'identity')."""
    if cfg['text_net'].get('name', 'transformer') == 'identity':
        return text_identity(sd, cfg['text_net'], tokens, token_masks)
    return text_transformer(sd, cfg['text_net'], tokens, token_masks)


def gated_video_input(cfg, vid, shallow_vid, vid_masks, correl, b, allow_sfonly=True):
    """model.py:527-551 for query ``b``: returns (x (1, C_in, T), mask (1, T)).  PtTransformer (model.py:123-129) has
    the same lines without the sfonly branch (``allow_sfonly=False``)."""
    vid = vid.clone()
    masks = vid_masks.clone()
    vid_len = int(masks.sum())
    weight = topk_block_gate(correl[b], vid_len, cfg['sn'], cfg['sratio'])
    all_weight = torch.zeros_like(masks)
    all_weight[0, :vid_len] = weight
    vid = vid * all_weight.unsqueeze(1)
    if not cfg['msf']:
        masks = torch.logical_and(all_weight.bool(), masks)
    elif allow_sfonly and cfg.get('sfonly', False):
        vid = shallow_vid
    else:
        vid = torch.cat([vid, shallow_vid], dim=1)
    if cfg.get('scat', False):
        vid = torch.cat([vid, correl[b][None, None, :]], dim=1)
    return vid, masks


def forward_eval(sd: SD, cfg, vid: Tensor, shallow_vid: Tensor, vid_masks: Tensor,
                 text: Sequence[Tensor], text_cls: Tensor, text_masks: Sequence[Tensor],
                 return_intermediates: bool = False, second_fusion: bool = False):
    """PtTransformerEarlyFusionIterative._drop_forward_eval, model.py:480-565.

    vid, shallow_vid (1, D, T); vid_masks (1, T) bool; text: NQ encoded texts (1,TE,Lk);
    text_cls (NQ, D); text_masks: NQ x (1,1,Lk).  Returns three lists (len NQ) of tuples
    (len L): logits (1,T_l), offsets (1,T_l,2), masks (1,T_l).
    ``cfg`` is the ``opt.model`` mapping.
    """
    assert vid.size(0) == 1
    correl = sidekick_scores(shallow_vid, text_cls, cfg['norm'])
    logits_list, offsets_list, masks_list, inter = [], [], [], []
    for b, (txt, txt_mask) in enumerate(zip(text, text_masks)):
        x, masks = gated_video_input(cfg, vid, shallow_vid, vid_masks, correl, b)
        m = masks.unsqueeze(1)
        x, m = masked_conv1d(x, m, sd['vid_map.conv.weight'], sd['vid_map.conv.bias'])
        fused, fm = xattn_fusion(sd, cfg['fusion'], x, m, txt, txt_mask)
        fpn, fpn_masks = video_transformer(sd, cfg['vid_net'], fused, fm)
        l1, l2, off, om = fuse_and_predict(sd, cfg, fpn, fpn_masks, txt, txt_mask, second_fusion)
        logits_list.append(l2)
        offsets_list.append(off)
        masks_list.append(om)
        if return_intermediates:
            inter.append(dict(vid_map=x, fused=fused, fpn=fpn, logits1=l1))
    if return_intermediates:
        return logits_list, offsets_list, masks_list, dict(correl=correl, per_query=inter)
    return logits_list, offsets_list, masks_list


def forward_eval_early_fusion(sd: SD, cfg, vid: Tensor, shallow_vid: Tensor, vid_masks: Tensor,
                              text: Sequence[Tensor], text_cls: Tensor, text_masks: Sequence[Tensor], second_fusion: bool = True):
    """PtTransformerEarlyFusion._drop_forward_eval, model.py:217-288: the iterative model's path (vid_map, fusion, vid_net,
    optional second fusion) with cls_head / reg_head directly on the pyramid (fuse_and_predict, model.py:204-209)."""
    assert vid.size(0) == 1
    correl = sidekick_scores(shallow_vid, text_cls, cfg['norm'])
    logits_list, offsets_list, masks_list = [], [], []
    for b, (txt, txt_mask) in enumerate(zip(text, text_masks)):
        x, masks = gated_video_input(cfg, vid, shallow_vid, vid_masks, correl, b, allow_sfonly=False)
        m = masks.unsqueeze(1)
        x, m = masked_conv1d(x, m, sd['vid_map.conv.weight'], sd['vid_map.conv.bias'])
        fused, fm = xattn_fusion(sd, cfg['fusion'], x, m, txt, txt_mask)
        fpn, fpn_masks = video_transformer(sd, cfg['vid_net'], fused, fm)
        if second_fusion:
            fpn, fpn_masks = xattn_fusion_pyramid(sd, cfg['fusion'], fpn, fpn_masks, txt, txt_mask)
        logits, _ = cls_head(sd, 'cls_head', fpn, fpn_masks, cfg['cls_head'].get('n_layers', 2))
        offsets, out_masks = reg_head(sd, 'reg_head', fpn, fpn_masks, cfg['reg_head'].get('n_layers', 2))
        logits_list.append(logits)
        offsets_list.append(offsets)
        masks_list.append(out_masks)
    return logits_list, offsets_list, masks_list


def forward_train(sd: SD, cfg, vid: Tensor, shallow_vid: Tensor, vid_masks: Tensor, tokens: Tensor, token_masks: Tensor,
                  text_cls: Tensor, text_size: Sequence[int]):
    """PtTransformerEarlyFusionIterative._drop_forward (training mode), model.py:567-632, FORWARD VALUES with every dropout
    / drop-path probability 0 (their modules are then identities: blocks.py:670-684, model.py:421,614), including the
    Dropout(0.5) hard-coded into the refinement TCN (tcn.py:5,13; model.py:424-425).

    vid, shallow_vid (bs, D, T); vid_masks (bs, T); tokens (sum(text_size), C_t, Lq) raw text features with token_masks
    (sum(text_size), 1, Lq); text_cls (sum(text_size), D); text_size[b] queries of video b.  The reference repeats video b
    text_size[b] times (:579-582), gates every row with its own query's scores (:593-606), encodes the text batch (:624) and
    runs fusion / vid_net / fuse_and_predict on the (video, query) rows; no operation mixes rows, so the rows are computed
    one at a time here.  Returns (fpn_logits1, fpn_logits2, fpn_offsets, fpn_masks): tuples over the levels of
    (B', T_l), (B', T_l), (B', T_l, 2), (B', T_l) with B' = sum(text_size)."""
    n_levels = cfg['vid_net']['arch'][2]
    rows = [[] for _ in range(4)]
    q = 0
    for b, k in enumerate(text_size):
        v, s, m = vid[b:b + 1], shallow_vid[b:b + 1], vid_masks[b:b + 1]
        texts, tmasks = [], []
        for i in range(q, q + k):
            t, tm = encode_text(sd, cfg, tokens[i:i + 1], token_masks[i:i + 1])
            texts.append(t)
            tmasks.append(tm)
        l2, off, om, inter = forward_eval(sd, cfg, v, s, m, texts, text_cls[q:q + k], tmasks, return_intermediates=True)
        for i in range(k):
            rows[0].append(inter['per_query'][i]['logits1'])
            rows[1].append(l2[i])
            rows[2].append(off[i])
            rows[3].append(om[i])
        q += k
    return tuple(tuple(torch.cat([r[l] for r in part], 0) for l in range(n_levels)) for part in rows)


# ----------------------------------------------------------------------------------
# point losses (libs/modeling/loss.py), forward values
# ----------------------------------------------------------------------------------
def sigmoid_focal_loss(inputs: Tensor, targets: Tensor, alpha: float = -1, gamma: float = 2.0, smoothing: bool = True) -> Tensor:
    """loss.py:5-57, reduction 'none'."""
    x, t = inputs.float(), targets.float()
    pos = (t >= 0.5).float()
    p = 1.0 / (1.0 + torch.exp(-x))
    p_t = p * t + (1 - p) * (1 - t) if smoothing else p * pos + (1 - p) * (1 - pos)
    ce = (1 - t) * x + torch.clamp(-x, min=0) + torch.log1p(torch.exp(-x.abs()))      # BCE with logits, overflow-free form
    loss = ce * (1 - p_t) ** gamma
    if alpha >= 0:
        loss = (alpha * pos + (1 - alpha) * (1 - pos)) * loss
    return loss


def ctr_iou_loss(input_offsets: Tensor, target_offsets: Tensor, kind: str = 'diou', eps: float = 1e-8) -> Tensor:
    """ctr_giou_loss (loss.py:60-109; reduces to 1 - IoU) / ctr_diou_loss (loss.py:111-166), reduction 'none'."""
    lp, rp = input_offsets[:, 0].float(), input_offsets[:, 1].float()
    lg, rg = target_offsets[:, 0].float(), target_offsets[:, 1].float()
    inter = torch.minimum(rp, rg) + torch.minimum(lp, lg)
    union = (lp + rp) + (lg + rg) - inter
    loss = 1.0 - inter / union.clamp(min=eps)
    if kind == 'diou':
        len_c = torch.maximum(lp, lg) + torch.maximum(rp, rg)
        rho = 0.5 * (rp - lp - rg + lg)
        loss = loss + (rho / len_c.clamp(min=eps)) ** 2
    return loss


def forward_eval_late_fusion(sd: SD, cfg, vid: Tensor, shallow_vid: Tensor, vid_masks: Tensor,
                             text: Sequence[Tensor], text_cls: Tensor, text_masks: Sequence[Tensor]):
    """PtTransformer._drop_forward with eval=True (late fusion), model.py:83-161: the gated, concatenated features go
    straight into vid_net (in_dim = 2D); the fusion stack is applied to every pyramid level; cls_head / reg_head only."""
    assert vid.size(0) == 1
    correl = sidekick_scores(shallow_vid, text_cls, cfg['norm'])
    vn = dict(cfg['vid_net'])
    logits_list, offsets_list, masks_list = [], [], []
    for b, (txt, txt_mask) in enumerate(zip(text, text_masks)):
        x, masks = gated_video_input(cfg, vid, shallow_vid, vid_masks, correl, b, allow_sfonly=False)
        fpn, fpn_masks = video_transformer(sd, vn, x, masks)
        fpn, fpn_masks = xattn_fusion_pyramid(sd, cfg['fusion'], fpn, fpn_masks, txt, txt_mask)
        lg, _ = cls_head(sd, 'cls_head', fpn, fpn_masks, cfg['cls_head'].get('n_layers', 2))
        off, om = reg_head(sd, 'reg_head', fpn, fpn_masks, cfg['reg_head'].get('n_layers', 2))
        logits_list.append(lg)
        offsets_list.append(off)
        masks_list.append(om)
    return logits_list, offsets_list, masks_list


def forward_train_single_head(sd: SD, cfg, kind: str, vid: Tensor, shallow_vid: Tensor, vid_masks: Tensor, tokens: Tensor, token_masks: Tensor,
                              text_cls: Tensor, text_size: Sequence[int]):
    """The training-mode forward (eval=False, every dropout probability 0) of the two classes with one classification head:
    kind 'late' = PtTransformer._drop_forward (model.py:83-147), 'early2' / 'early1' = PtTransformerEarlyFusion._drop_forward with /
    without the second fusion (model.py:320-362).  As in forward_train: video b is repeated text_size[b] times, every row is gated by
    its own query's scores, the text batch is encoded (model.py:140-147, :349-362) and nothing mixes rows -- the rows are computed one
    at a time with the evaluation restatements.  Returns (fpn_logits, fpn_offsets, fpn_masks): tuples over the levels of (B', T_l),
    (B', T_l, 2), (B', T_l)."""
    n_levels = cfg['vid_net']['arch'][2]
    rows = [[] for _ in range(3)]
    q = 0
    for b, k in enumerate(text_size):
        v, s, m = vid[b:b + 1], shallow_vid[b:b + 1], vid_masks[b:b + 1]
        texts, tmasks = [], []
        for i in range(q, q + k):
            t, tm = encode_text(sd, cfg, tokens[i:i + 1], token_masks[i:i + 1])
            texts.append(t)
            tmasks.append(tm)
        if kind == 'late':
            lg, off, om = forward_eval_late_fusion(sd, cfg, v, s, m, texts, text_cls[q:q + k], tmasks)
        else:
            lg, off, om = forward_eval_early_fusion(sd, cfg, v, s, m, texts, text_cls[q:q + k], tmasks, second_fusion=kind == 'early2')
        for i in range(k):
            rows[0].append(lg[i])
            rows[1].append(off[i])
            rows[2].append(om[i])
        q += k
    return tuple(tuple(torch.cat([r[l] for r in part], 0) for l in range(n_levels)) for part in rows)


def forward_eval_window(sd: SD, cfg, vid_w: Tensor, shallow_w: Tensor, mask_w: Tensor, text, text_masks, gate_w: Tensor,
                        pe_w=None):
    """The eval forward on a window of a longer video with an externally selected gate (NQ, Tw) and the
    window's position-encoding slice (E, Tw).  Mirrors model.py:543-563 with the gate given."""
    logits_list, offsets_list, masks_list = [], [], []
    for b, (txt, txt_mask) in enumerate(zip(text, text_masks)):
        g = gate_w[b][None, None, :]
        x = vid_w * g
        masks = mask_w.clone()
        if not cfg['msf']:
            masks = torch.logical_and(gate_w[b][None].bool(), masks)
        else:
            x = torch.cat([x, shallow_w], dim=1)
        m = masks.unsqueeze(1)
        x, m = masked_conv1d(x, m, sd['vid_map.conv.weight'], sd['vid_map.conv.bias'])
        fused, fm = xattn_fusion(sd, cfg['fusion'], x, m, txt, txt_mask)
        fpn, fpn_masks = video_transformer(sd, cfg['vid_net'], fused, fm, pe_override=pe_w)
        _, l2, off, om = fuse_and_predict(sd, cfg, fpn, fpn_masks)
        logits_list.append(l2)
        offsets_list.append(off)
        masks_list.append(om)
    return logits_list, offsets_list, masks_list


# ----------------------------------------------------------------------------------
# point generator + post-processing (Evaluator)
# ----------------------------------------------------------------------------------
def generate_points(max_seq_len: int, num_fpn_levels: int, regression_range: float = 4,
                    sigma: float = 1, use_offset: bool = False) -> List[Tensor]:
    """PtGenerator model.py:668-743: per level (T_l, 4) = [coord, range lo, range hi, stride]."""
    assert max_seq_len % 2 ** (num_fpn_levels - 1) == 0
    ranges = [(0, regression_range)]
    rr = regression_range
    for l in range(1, num_fpn_levels):
        lo, hi = rr * sigma, rr * 2
        if l == num_fpn_levels - 1:
            hi = max(hi, max_seq_len + 1)
        ranges.append((lo, hi))
        rr = hi
    tics = torch.arange(0, max_seq_len, 1.0)
    pts = []
    for l in range(num_fpn_levels):
        stride = 2 ** l
        c = tics[::stride][:, None].clone()
        if use_offset:
            c += 0.5 * stride
        r = torch.as_tensor(ranges[l], dtype=torch.float32)[None].repeat(len(c), 1)
        s = torch.full((len(c), 1), float(stride))
        pts.append(torch.cat((c, r, s), 1))
    return pts


def padded_length(vid_len: int, max_vid_len: int, num_fpn_levels: int, mha_win_size: int,
                  vid_stride: int = 1) -> int:
    """Evaluator pad rule worker_v2.py:769-781,969-976."""
    min_chunk = 1
    for l in range(num_fpn_levels):
        s = 2 ** l
        if mha_win_size > 0:
            s *= (mha_win_size // 2) * 2
        min_chunk = max(min_chunk, s)
    input_len = max_vid_len * vid_stride
    if vid_len > input_len:
        stride = min_chunk * vid_stride
        input_len = (vid_len + (stride - 1)) // stride * stride
    return input_len


def collect_segments(fpn_points, fpn_logits, fpn_offsets, fpn_masks, pre_nms_thresh: float = 0.001,
                     pre_nms_topk: int = 2000, seg_len_thresh: float = 0.1, stable: bool = True, ext_scores=None):
    """Evaluator._collect_segments worker_v2.py:1131-1187; ``ext_scores`` (T,) as worker_v2.py:1150-1156.
    ``stable=True`` pins the (reference-unspecified) argsort tie order to lowest index
    first, which is what the HIP path implements."""
    pts_l, sc_l, off_l = [], [], []
    for points, logits, offsets, masks in zip(fpn_points, fpn_logits, fpn_offsets, fpn_masks):
        logits, offsets, masks = logits[0], offsets[0], masks[0]
        scores = torch.sigmoid(logits)
        if ext_scores is not None:
            scores = scores * ext_scores
            ext_scores = F.max_pool1d(ext_scores[None, None], kernel_size=3, stride=2, padding=1)[0, 0]
        scores = scores * masks.float()
        keep = scores > pre_nms_thresh
        pts_l.append(points[keep])
        sc_l.append(scores[keep])
        off_l.append(offsets[keep])
    points, scores, offsets = torch.cat(pts_l), torch.cat(sc_l), torch.cat(off_l)
    n_topk = min(len(points), pre_nms_topk)
    idx = scores.argsort(descending=True, stable=stable)[:n_topk]
    points, scores, offsets = points[idx], scores[idx], offsets[idx]
    ctr = points[:, 0]
    left = ctr - offsets[:, 0] * points[:, 3]
    right = ctr + offsets[:, 1] * points[:, 3]
    segs = torch.stack((left, right), dim=-1)
    keep = (right - left) > seg_len_thresh
    return segs[keep], scores[keep]


def segment_voting(nms_segs, all_segs, all_scores, iou_thresh):
    """libs/nms/nms.py:64-103."""
    a = nms_segs[:, None]
    b = all_segs[None, :]
    left = torch.maximum(a[..., 0], b[..., 0])
    right = torch.minimum(a[..., 1], b[..., 1])
    overlap = (right - left).clamp(min=0)
    union = (a[..., 1] - a[..., 0]) + (b[..., 1] - b[..., 0]) - overlap
    iou = overlap / union
    w = (iou >= iou_thresh).float() * all_scores[None]
    w = w / w.sum(dim=1, keepdim=True)
    return w @ all_segs


def batched_nms(segs, scores, iou_thresh, min_score, max_num_segs, mode='soft_nms', sigma=0.5,
                voting_thresh=0.75, nms_fn=None, softnms_fn=None):
    """libs/nms/nms.py:106-148 with the native calls injected (``nms_fn``/``softnms_fn``
    default to the C oracle in oracle/nms_ref.c through oracle/nms_oracle.py)."""
    if nms_fn is None or softnms_fn is None:
        from . import nms_oracle
        nms_fn = nms_fn or nms_oracle.nms
        softnms_fn = softnms_fn or nms_oracle.softnms
    if len(segs) == 0:
        return torch.zeros(0, 2), torch.zeros(0)
    if mode is not None:
        if mode == 'nms':
            s, c = segs, scores
            if min_score > 0:
                keep = c > min_score
                s, c = s[keep], c[keep]
            idx = nms_fn(s.contiguous(), c.contiguous(), float(iou_thresh))
            if max_num_segs > 0:
                idx = idx[:min(max_num_segs, len(idx))]
            nms_segs, nms_scores = s[idx].contiguous(), c[idx].contiguous()
        elif mode == 'soft_nms':
            out = segs.new_empty((len(segs), 3))
            idx = softnms_fn(segs.contiguous(), scores.contiguous(), out, float(iou_thresh),
                             float(sigma), float(min_score), 2)
            n = len(idx)
            if max_num_segs > 0:
                n = min(n, max_num_segs)
            nms_segs, nms_scores = out[:n, :2].contiguous(), out[:n, 2].contiguous()
        else:
            raise NotImplementedError('invalid NMS mode')
        if voting_thresh > 0:
            nms_segs = segment_voting(nms_segs, segs, scores, voting_thresh)
    else:
        nms_segs, nms_scores = segs, scores
    idx = nms_scores.argsort(descending=True, stable=True)
    k = min(max_num_segs, len(nms_segs))
    return nms_segs[idx[:k]], nms_scores[idx[:k]]


def to_seconds(segs, vid_stride, clip_stride, clip_size, fps, duration):
    """worker_v2.py:1114-1122."""
    if len(segs) == 0:
        return segs
    segs = segs * vid_stride
    segs = (segs * clip_stride + 0.5 * clip_size) / fps
    return torch.clamp(segs, min=0, max=duration)
