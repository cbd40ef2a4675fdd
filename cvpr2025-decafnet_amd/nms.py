"""1-D NMS front-ends on the HIP kernels (csrc/postproc.hip) with the reference's signatures.

* ``nms`` / ``softnms``  == the functions of the extension module ``nms_1d_cpu_vg``
  (libs/nms/src/nms_cpu.cpp:184-194): CPU tensors in, CPU ``int64`` indices out, ``dets``
  filled in place, same checks and error messages.  The repository-root package
  ``nms_1d_cpu_vg`` re-exports them under the reference's module name.
* ``batched_nms``        == libs/nms/nms.py:106-148 (same arguments and defaults); device
  tensors stay on the device, results come back on the input's device.
* ``collect_segments``   == Evaluator._collect_segments (libs/worker_v2.py:1131-1187) for all
  queries of a video at once.

Ties between equal scores are broken by the lower index (the reference leaves them to
at::sort / argsort, which is unspecified).
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib

NMS_CAPACITY = 4096        # candidates per query that stay in one workgroup's LDS; larger problems run the same kernels over
                           # a global-memory scratch block (any n, like the reference: nms_cpu.cpp:20-63)


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError('nms_1d: no MI355X visible and there is no CPU fallback in this package')
    return torch.device('cuda', torch.cuda.current_device())


def _check_cpu_input(x, name):
    # CHECK_CPU_INPUT, nms_cpu.cpp:11-17
    if x.is_cuda:
        raise RuntimeError(f'{name} must be a CPU tensor')
    if not x.is_contiguous():
        raise RuntimeError(f'{name} must be contiguous')


def _check_float(x, name):
    if x.dtype != torch.float32:
        raise RuntimeError(f'expected scalar type Float but found {str(x.dtype).replace("torch.", "").capitalize()} ({name})')


def nms_device(segs, scores, counts, n_max, stride, iou_thresh):
    """Batched hard NMS on device tensors: segs (nq, stride, 2), scores (nq, stride), counts (nq) int32 or None."""
    lib = _lib.lib()
    nq = segs.shape[0]
    keep = torch.empty(nq, stride, dtype=torch.int64, device=segs.device)
    kc = torch.empty(nq, dtype=torch.int32, device=segs.device)
    _lib.check(lib.dcf_nms_1d(_lib.ptr(segs), _lib.ptr(scores), _lib.ptr(counts), nq, n_max, stride, float(iou_thresh),
                              _lib.ptr(keep), _lib.ptr(kc), _lib.current_stream()), 'dcf_nms_1d')
    return keep, kc


def softnms_device(segs, scores, counts, n_max, stride, iou_thresh, sigma, min_score, method, max_iters=0):
    lib = _lib.lib()
    nq = segs.shape[0]
    dets = torch.empty(nq, stride, 3, dtype=torch.float32, device=segs.device)
    inds = torch.empty(nq, stride, dtype=torch.int64, device=segs.device)
    oc = torch.empty(nq, dtype=torch.int32, device=segs.device)
    _lib.check(lib.dcf_softnms_1d(_lib.ptr(segs), _lib.ptr(scores), _lib.ptr(counts), nq, n_max, stride, float(iou_thresh),
                                  float(sigma), float(min_score), int(method), int(max_iters), _lib.ptr(dets),
                                  _lib.ptr(inds), _lib.ptr(oc), _lib.current_stream()), 'dcf_softnms_1d')
    return dets, inds, oc


def voting_device(nms_segs, n1_counts, n1_max, all_segs, all_scores, n2_counts, n2_max, iou_thresh):
    """nms_segs (nq, n1_stride, ld>=2); all_segs (nq, n2_stride, 2); returns (nq, n1_stride, 2)."""
    lib = _lib.lib()
    nq, n1_stride, ld = nms_segs.shape
    out = torch.zeros(nq, n1_stride, 2, dtype=torch.float32, device=nms_segs.device)
    _lib.check(lib.dcf_segment_voting(_lib.ptr(nms_segs), ld, _lib.ptr(n1_counts), n1_max, n1_stride, _lib.ptr(all_segs),
                                      _lib.ptr(all_scores), _lib.ptr(n2_counts), n2_max, all_segs.shape[1],
                                      float(iou_thresh), nq, _lib.ptr(out), _lib.current_stream()), 'dcf_segment_voting')
    return out


# ---------------------------------------------------------------------------------------------
# extension-module ABI (CPU tensors)
# ---------------------------------------------------------------------------------------------
def nms(segs, scores, iou_thresh):
    """nms(segs: Tensor[n,2] f32 CPU, scores: Tensor[n] CPU, iou_thresh: float) -> Tensor[k] int64."""
    _check_cpu_input(segs, 'segs')
    _check_cpu_input(scores, 'scores')
    if segs.numel() == 0:
        return torch.empty(0, dtype=torch.int64)
    _check_float(segs, 'segs')
    n = segs.shape[0]
    dev = _device()
    keep, kc = nms_device(segs.to(dev)[None], scores.to(dev, torch.float32)[None].contiguous(), None, n, n, iou_thresh)
    k = int(kc.item())
    return keep[0, :k].cpu()


def softnms(segs, scores, dets, iou_thresh, sigma, min_score, method):
    """softnms(segs, scores, dets (n,3) out-param, iou_thresh, sigma, min_score, method) -> Tensor[k] int64."""
    _check_cpu_input(segs, 'segs')
    _check_cpu_input(scores, 'scores')
    _check_cpu_input(dets, 'dets')
    if segs.numel() == 0:
        return torch.empty(0, dtype=torch.int64)
    for x, nme in ((segs, 'segs'), (scores, 'scores'), (dets, 'dets')):
        _check_float(x, nme)
    n = segs.shape[0]
    dev = _device()
    d, inds, oc = softnms_device(segs.to(dev)[None], scores.to(dev)[None], None, n, n, iou_thresh, sigma, min_score, method)
    k = int(oc.item())
    # the reference writes one dets row per PICK; with pruning there are exactly k picks
    dets[:k].copy_(d[0, :k].cpu())
    return inds[0, :k].cpu()


# ---------------------------------------------------------------------------------------------
# libs/nms/nms.py
# ---------------------------------------------------------------------------------------------
def segment_voting(nms_segs, all_segs, all_scores, iou_thresh):
    """libs/nms/nms.py:64-103."""
    dev = all_segs.device if all_segs.is_cuda else _device()
    out = voting_device(nms_segs.to(dev, torch.float32).contiguous()[None], None, nms_segs.shape[0],
                        all_segs.to(dev, torch.float32).contiguous()[None], all_scores.to(dev, torch.float32).contiguous()[None],
                        None, all_segs.shape[0], iou_thresh)[0]
    return out.to(nms_segs.device)


def batched_nms(segs, scores, iou_thresh, min_score, max_num_segs, mode='soft_nms', sigma=0.5, voting_thresh=0.75):
    """libs/nms/nms.py:106-148.  segs (n,2), scores (n,) on any device."""
    if len(segs) == 0:
        return torch.zeros(0, 2), torch.zeros(0)
    in_dev = segs.device
    dev = in_dev if segs.is_cuda else _device()
    segs_d = segs.to(dev, torch.float32).contiguous()
    scores_d = scores.to(dev, torch.float32).contiguous()
    if mode is not None:
        if mode == 'nms':
            s, c = segs_d, scores_d
            if min_score > 0:                                  # NMSop.forward, nms.py:13-16
                keep = c > min_score
                s, c = s[keep].contiguous(), c[keep].contiguous()
            n = s.shape[0]
            if n == 0:
                nms_segs, nms_scores = s, c
            else:
                idx, kc = nms_device(s[None], c[None], None, n, n, iou_thresh)
                k = int(kc.item())
                if max_num_segs > 0:
                    k = min(k, max_num_segs)
                idx = idx[0, :k]
                nms_segs, nms_scores = s[idx].contiguous(), c[idx].contiguous()
        elif mode == 'soft_nms':
            n = segs_d.shape[0]
            # SoftNMSop only keeps the first max_num_segs picks (nms.py:54-59): stop there
            d, inds, oc = softnms_device(segs_d[None], scores_d[None], None, n, n, iou_thresh, sigma, min_score, 2,
                                         max_iters=max_num_segs if max_num_segs > 0 else 0)
            k = int(oc.item())
            if max_num_segs > 0:
                k = min(k, max_num_segs)
            nms_segs, nms_scores = d[0, :k, :2].contiguous(), d[0, :k, 2].contiguous()
        else:
            raise NotImplementedError('invalid NMS mode')
        if voting_thresh > 0 and len(nms_segs) > 0:
            nms_segs = voting_device(nms_segs[None].contiguous(), None, nms_segs.shape[0], segs_d[None], scores_d[None],
                                     None, segs_d.shape[0], voting_thresh)[0]
    else:
        nms_segs, nms_scores = segs_d, scores_d
    idx = nms_scores.argsort(descending=True, stable=True)
    k = min(max_num_segs, len(nms_segs))
    return nms_segs[idx[:k]].to(in_dev), nms_scores[idx[:k]].to(in_dev)


def batched_nms_queries(segs, scores, counts, iou_thresh, min_score, max_num_segs, mode='soft_nms', sigma=0.5, voting_thresh=0.75):
    """``batched_nms`` (libs/nms/nms.py:106-148) for ALL queries of a video in one pass on the device, without a single
    host synchronisation: segs (nq, K, 2), scores (nq, K), counts (nq) int32 as ``collect_segments`` returns them (every
    row sorted by descending score, which is what lets NMSop's ``scores > min_score`` filter, nms.py:13-16, be a count).
    Returns device tensors out_segs (nq, max_num_segs, 2), out_scores (nq, max_num_segs), out_counts (nq) int32; rows
    beyond out_counts[q] are padding.  Same values per query as ``batched_nms``."""
    assert segs.is_cuda and segs.dim() == 3 and max_num_segs > 0
    nq, K = scores.shape
    M = int(max_num_segs)
    dev = segs.device
    segs = segs.contiguous()
    scores = scores.contiguous()
    counts = counts.to(torch.int32).contiguous()
    if K == 0:
        return segs.new_zeros(nq, M, 2), scores.new_zeros(nq, M), counts.new_zeros(nq)
    if mode == 'soft_nms':
        d, _, oc = softnms_device(segs, scores, counts, K, K, iou_thresh, sigma, min_score, 2, max_iters=M)
        kc = torch.clamp(oc, max=M)
        top = d[:, :M].contiguous()                                       # (nq, M', 3) picks in pick order
        nms_segs, nms_scores = top[..., :2], top[..., 2]
    elif mode == 'nms':
        live = torch.arange(K, device=dev)[None] < counts[:, None]
        c_eff = ((scores > min_score) & live).sum(1).to(torch.int32) if min_score > 0 else counts
        idx, kc = nms_device(segs, scores, c_eff, K, K, iou_thresh)
        kc = torch.clamp(kc, max=M)
        idx = idx[:, :M].clamp_(0, K - 1)                                 # entries beyond kc are undefined: keep the gather in range
        nms_segs = torch.gather(segs, 1, idx[..., None].expand(-1, -1, 2))
        nms_scores = torch.gather(scores, 1, idx)
        top = nms_segs
    elif mode is None:
        kc = torch.clamp(counts, max=M)
        nms_segs, nms_scores, top = segs[:, :M], scores[:, :M], segs[:, :M].contiguous()
    else:
        raise NotImplementedError('invalid NMS mode')
    m_ = nms_scores.shape[1]
    if m_ < M:                                                            # fewer candidates than max_num_segs
        pad = M - m_
        nms_segs = torch.nn.functional.pad(nms_segs, (0, 0, 0, pad))
        nms_scores = torch.nn.functional.pad(nms_scores, (0, pad))
        top = torch.nn.functional.pad(top, (0, 0, 0, pad))
    if mode is not None and voting_thresh > 0:
        nms_segs = voting_device(top.contiguous(), kc, M, segs, scores, counts, K, voting_thresh)
    if mode is not None:
        # batched_nms ends with a descending argsort of the kept scores (nms.py:143-146).  Both NMS variants already emit their
        # picks in non-increasing score order (greedy NMS walks the sorted candidates; every soft-NMS pick is the maximum of a
        # set that only shrinks and decays), and the reference's sort is stable, so it is the identity here: no launches.
        return nms_segs, nms_scores, kc
    valid = torch.arange(M, device=dev)[None] < kc[:, None]
    order = torch.where(valid, nms_scores, nms_scores.new_full((), float('-inf'))).sort(dim=1, descending=True, stable=True)[1]
    out_segs = torch.gather(nms_segs, 1, order[..., None].expand(-1, -1, 2))
    out_scores = torch.gather(nms_scores, 1, order)
    return out_segs, out_scores, kc


def collect_segments(logits, offsets, masks, T, n_levels, pre_nms_thresh=0.001, pre_nms_topk=2000, seg_len_thresh=0.1,
                     ext_scores=None):
    """Evaluator._collect_segments for all queries: logits (nq,S), offsets (nq,S,2), masks (nq,S) on the GPU ->
    segs (nq, topk, 2), scores (nq, topk), counts (nq) int32 (device; rows beyond counts[q] are undefined).
    ``ext_scores`` (nq, T) or (T,) are the optional external per-clip scores of worker_v2.py:964-967,1150-1156."""
    lib = _lib.lib()
    nq = logits.shape[0]
    dev = logits.device
    ext = None
    if ext_scores is not None:
        ext = ext_scores.to(device=dev, dtype=torch.float32)
        ext = (ext[None].expand(nq, -1) if ext.dim() == 1 else ext).contiguous()
        assert ext.shape == (nq, T), (ext.shape, nq, T)
    segs = torch.empty(nq, pre_nms_topk, 2, dtype=torch.float32, device=dev)
    scores = torch.empty(nq, pre_nms_topk, dtype=torch.float32, device=dev)
    counts = torch.empty(nq, dtype=torch.int32, device=dev)
    _lib.check(lib.dcf_collect_segments_ext(_lib.ptr(logits.contiguous()), _lib.ptr(offsets.contiguous()),
                                            _lib.ptr(masks.contiguous()), _lib.ptr(ext) if ext is not None else None,
                                            nq, T, n_levels, float(pre_nms_thresh),
                                            int(pre_nms_topk), float(seg_len_thresh), _lib.ptr(segs), _lib.ptr(scores),
                                            _lib.ptr(counts), _lib.current_stream()), 'dcf_collect_segments_ext')
    return segs, scores, counts
