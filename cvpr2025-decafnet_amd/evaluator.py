"""Hot-path harness: the parts of the reference ``Evaluator`` that sit on the grounding path
(libs/worker_v2.py:726-1187), on device tensors, for all queries of a video at once.

    padded_length      worker_v2.py:769-781, 969-976
    GroundingEvaluator.predict   _forward (930-1026) + _generate_proposals (1063-1129):
        pad -> encode_text -> model(..., eval=True) -> collect_segments -> batched NMS -> seconds

The metric loop (R@k / IoU counting, worker_v2.py:857-910) and data loading are out of scope.
"""
from __future__ import annotations

import time
from collections import defaultdict

import torch
import torch.nn.functional as F

from . import nms as _nms


def min_chunk_size(num_fpn_levels: int, mha_win_size: int) -> int:
    """worker_v2.py:769-781"""
    m = 1
    for l in range(num_fpn_levels):
        s = 2 ** l
        if mha_win_size > 0:
            s *= (mha_win_size // 2) * 2
        m = max(m, s)
    return m


def padded_length(vid_len: int, max_vid_len: int, num_fpn_levels: int, mha_win_size: int, vid_stride: int = 1) -> int:
    """worker_v2.py:969-976: short videos pad to max_vid_len, long ones to the next multiple of the chunk size."""
    input_len = max_vid_len * vid_stride
    if vid_len > input_len:
        stride = min_chunk_size(num_fpn_levels, mha_win_size) * vid_stride
        input_len = (vid_len + (stride - 1)) // stride * stride
    return input_len


class GroundingEvaluator:
    """``opt`` needs ``opt.model.{max_vid_len, num_fpn_levels, mha_win_size, vid_stride}``, ``opt.eval.{pre_nms_topk,
    pre_nms_thresh, seg_len_thresh}`` and ``opt.nms`` (libs/core/opt.py:174-194)."""

    def __init__(self, opt, model):
        self.opt, self.model = opt, model
        mo = opt['model']
        self.max_vid_len = mo['max_vid_len']
        self.vid_stride = mo.get('vid_stride', 1)
        self.num_fpn_levels = mo['num_fpn_levels']
        self.mha_win_size = mo['mha_win_size']
        assert self.max_vid_len % min_chunk_size(self.num_fpn_levels, self.mha_win_size) == 0, \
            'max video length must be a multiple of the chunk size'            # worker_v2.py:778-780
        ev = opt['eval']
        self.pre_nms_topk, self.pre_nms_thresh, self.seg_len_thresh = ev['pre_nms_topk'], ev['pre_nms_thresh'], ev['seg_len_thresh']
        self.nms_cfg = dict(opt['nms'])
        self.time_dict = defaultdict(list)

    @torch.no_grad()
    def forward(self, data):
        """data: vid (D,T), shallow_vid (D,T), text: tuple of (C_t, Lq) token tensors, text_cls (NQ,D).
        Returns the flat device outputs (logits (NQ,S), offsets (NQ,S,2), masks (NQ,S)) and T_padded."""
        dev = next(self.model.parameters()).device
        t0 = time.perf_counter()
        tokens = data['text'] if isinstance(data['text'], (tuple, list)) else (data['text'],)
        texts, tmasks = [], []
        for tok in tokens:
            tok = tok[None].to(dev, non_blocking=True)
            m = torch.ones(1, 1, tok.size(-1), dtype=torch.bool, device=dev)
            t, m = self.model.encode_text(tok, m)
            texts.append(t)
            tmasks.append(m)
        vid, shallow = data['vid'], data['shallow_vid']
        vid_len = vid.size(-1)
        T = padded_length(vid_len, self.max_vid_len, self.num_fpn_levels, self.mha_win_size, self.vid_stride)
        window = F.pad(vid, (0, T - vid_len))[None].to(dev, non_blocking=True)
        shallow_window = F.pad(shallow, (0, T - vid_len))[None].to(dev, non_blocking=True)
        mask = (torch.arange(T, device=dev).view(1, -1) < vid_len)
        text_cls = data['text_cls'].to(dev, non_blocking=True)
        self.time_dict['prepare'].append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        out = self.model(window, shallow_window, mask, tuple(texts), text_cls, tuple(tmasks), eval=True)
        self.time_dict['forward'].append(time.perf_counter() - t0)
        self.outputs = out
        return self.model._last_flat, T

    @torch.no_grad()
    def generate_proposals(self, flat, T, data=None):
        """_collect_segments + batched_nms + seconds for every query; results as in worker_v2.py:1124."""
        logits, offsets, masks = flat
        t0 = time.perf_counter()
        segs, scores, counts = _nms.collect_segments(logits, offsets, masks, T, self.num_fpn_levels, self.pre_nms_thresh,
                                                     self.pre_nms_topk, self.seg_len_thresh)
        counts_h = counts.cpu()
        self.time_dict['post_process'].append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        results = []
        for q in range(logits.shape[0]):
            n = int(counts_h[q])
            s, c = _nms.batched_nms(segs[q, :n], scores[q, :n], **self.nms_cfg)
            if len(s) > 0 and data is not None:
                s = s * self.vid_stride
                s = (s * data['clip_stride'] + 0.5 * data['clip_size']) / data['fps']          # worker_v2.py:1120-1122
                s = torch.clamp(s, min=0, max=data['duration'])
            results.append({'segments': s, 'scores': c})
        self.time_dict['nms'].append(time.perf_counter() - t0)
        return results

    def predict(self, data):
        flat, T = self.forward(data)
        return self.generate_proposals(flat, T, data)
