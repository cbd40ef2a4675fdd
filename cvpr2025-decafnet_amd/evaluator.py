"""Hot-path harness: the parts of the reference ``Evaluator`` that sit on the grounding path
(libs/worker_v2.py:726-1187), on device tensors, for all queries of a video at once.

    padded_length      worker_v2.py:769-781, 969-976
    GroundingEvaluator.predict   _forward (930-1026) + _generate_proposals (1063-1129):
        pad -> encode_text -> model(..., eval=True) -> collect_segments -> batched NMS -> seconds

    iou, RecallCounter  libs/train_utils.py:81-96 and the metric loop of Evaluator.run, worker_v2.py:857-878, 890-901
    load_features       feature-file formats of libs/data/dataset.py:128-135, 398-399 ((T, C) on disk -> (C, T))
"""
from __future__ import annotations

import os
import time
from collections import defaultdict

import numpy as np
import torch
import torch.nn.functional as F

from . import nms as _nms


def iou(pred_segs, gt_segs):
    """Temporal IoU of (..., 2) [start, end] segments, broadcast over the leading dimensions.  Same value as the reference's
    ``iou`` (libs/train_utils.py:81-96): intersection clamped at 0 over (sum of lengths - intersection), no epsilon, so two
    empty segments give nan there as here."""
    lo = torch.maximum(pred_segs[..., 0], gt_segs[..., 0])
    hi = torch.minimum(pred_segs[..., 1], gt_segs[..., 1])
    inter = (hi - lo).clamp(min=0)
    total = (pred_segs[..., 1] - pred_segs[..., 0]) + (gt_segs[..., 1] - gt_segs[..., 0])
    return inter / (total - inter)


class RecallCounter:
    """Recall@k at temporal-IoU thresholds, the metric ``Evaluator.run`` accumulates (libs/worker_v2.py:857-878) and prints
    (:890-901).  A query is a hit at (k, t) when any of its k best-scored proposals overlaps the ground truth by IoU >= t;
    a query without proposals is a miss everywhere."""

    def __init__(self, ranks=(1, 5), iou_threshs=(0.3, 0.5)):
        self.ranks = tuple(int(k) for k in ranks)
        self.iou_threshs = np.asarray(iou_threshs, dtype=np.float64)
        self.hits = np.zeros((len(self.ranks), len(self.iou_threshs)))
        self.n_queries = 0

    # the reference's attribute names, for code that reads them
    @property
    def counts(self):
        return self.hits

    @property
    def text_cnt(self):
        return self.n_queries

    def update(self, results, targets):
        if len(results) != len(targets):
            raise ValueError(f'{len(results)} results for {len(targets)} ground-truth segments')
        for res, gt in zip(results, targets):
            segs, scores = res['segments'].cpu(), res['scores'].cpu()
            best_first = segs[torch.argsort(scores, descending=True)[:max(self.ranks)]]
            overlaps = iou(best_first, torch.as_tensor(gt, dtype=torch.float32)[None])
            for i, k in enumerate(self.ranks):
                best = float(overlaps[:k].max()) if len(overlaps[:k]) else 0.0
                self.hits[i] += best >= self.iou_threshs
        self.n_queries += len(targets)

    def metrics(self):
        return self.hits / max(self.n_queries, 1)

    def report(self):
        """the reference's table layout (worker_v2.py:890-901)"""
        m = self.metrics() * 100
        lines = ['', 'Final:']
        for i, k in enumerate(self.ranks):
            lines.append('-----')
            lines += [f'Rank@{k}, IoU@{t:.1f}: {m[i, j]:.2f}' for j, t in enumerate(self.iou_threshs)]
        return '\n'.join(lines + ['-----', ''])


def load_features(path_without_ext: str, fmt: str = 'npy'):
    """One feature file as the reference stores it ((T, C) on disk: npy / pt / pk0 / pk1 / pk_avg, libs/data/dataset.py:107-135)
    returned channel-major (C, T) like ``_load_vid_feats`` (:398-399).  Multi-source videos: ``data.load_video_features``."""
    from . import data as _data
    a = np.asarray(_data.load_clip_features(path_without_ext, fmt), dtype=np.float32)
    return torch.from_numpy(np.ascontiguousarray(a.transpose()))


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
        self._pinned = {}                              # free pinned host buffers of launch_proposals, by packed shape

    @classmethod
    def from_checkpoint(cls, opt, root=None, ckpt=None, device='cuda'):
        """What ``Evaluator.__init__`` + ``load_model`` do for the model (libs/worker_v2.py:747-749, 806-812): build it with
        ``create_model(opt)``, load ``<root>/models/<ckpt>.pth``['model_ema'] (a checkpoint written by the reference's
        ``Trainer``: the state_dict names are the parameter ABI), move it to the GPU, eval mode, no gradients.
        ``root`` / ``ckpt`` default to ``opt['_root']`` / ``opt['_ckpt']`` like the reference (eval.py:29-36)."""
        from . import modeling
        root = root if root is not None else opt['_root']
        ckpt = ckpt if ckpt is not None else opt['_ckpt']
        path = os.path.join(root, 'models', f'{ckpt}.pth')
        state = torch.load(path, map_location='cpu')
        if 'model_ema' not in state:
            raise KeyError(f"{path}: no 'model_ema' entry (keys: {sorted(state)})")
        model = modeling.create_model(opt)
        model.load_state_dict(state['model_ema'])
        model = model.to(device).eval().requires_grad_(False)
        return cls(opt, model)

    @torch.no_grad()
    def prepare(self, data, model=None):
        """pad -> encode_text (worker_v2.py:930-996): the argument tuple of ``model.forward`` for one video, its padded length
        and the padded external scores (or None)."""
        model_ = model or self.model
        dev = next(model_.parameters()).device
        t0 = time.perf_counter()
        tokens = data['text'] if isinstance(data['text'], (tuple, list)) else (data['text'],)
        texts, tmasks = [], []
        for tok in tokens:
            tok = tok[None].to(dev, non_blocking=True)
            m = torch.ones(1, 1, tok.size(-1), dtype=torch.bool, device=dev)
            t, m = model_.encode_text(tok, m)
            texts.append(t)
            tmasks.append(m)
        vid, shallow = data['vid'], data['shallow_vid']
        vid_len = vid.size(-1)
        T = padded_length(vid_len, self.max_vid_len, self.num_fpn_levels, self.mha_win_size, self.vid_stride)
        window = F.pad(vid, (0, T - vid_len))[None].to(dev, non_blocking=True)
        shallow_window = F.pad(shallow, (0, T - vid_len))[None].to(dev, non_blocking=True)
        mask = (torch.arange(T, device=dev).view(1, -1) < vid_len)
        text_cls = data['text_cls'].to(dev, non_blocking=True)
        # external per-clip scores (NQ, vid_len), zero-padded like the features (worker_v2.py:964-967,992-994)
        ext = data.get('ext_scores') if isinstance(data, dict) else None
        ext = None if ext is None else F.pad(ext.float(), (0, T - vid_len)).to(dev, non_blocking=True)
        self.time_dict['prepare'].append(time.perf_counter() - t0)
        return (window, shallow_window, mask, tuple(texts), text_cls, tuple(tmasks)), T, ext

    @torch.no_grad()
    def forward(self, data, model=None):
        """data: vid (D,T), shallow_vid (D,T), text: tuple of (C_t, Lq) token tensors, text_cls (NQ,D).
        Returns the flat device outputs (logits (NQ,S), offsets (NQ,S,2), masks (NQ,S)) and T_padded.
        ``model``: a ``self.model.replica()`` when several videos are kept in flight (``run(n_streams > 1)``)."""
        model_ = model or self.model
        args, T, self._window_ext = self.prepare(data, model_)
        t0 = time.perf_counter()
        out = model_(*args, eval=True)
        self.time_dict['forward'].append(time.perf_counter() - t0)
        self.outputs = out
        return model_._last_flat, T

    @torch.no_grad()
    def launch_proposals(self, flat, T, data=None, window_ext=None):
        """Enqueue _collect_segments + batched_nms for every query of a video and the ONE device -> host copy of the kept rows
        (<= max_num_segs segments + scores + count per query) into pinned memory, without waiting for any of it.  Returns a
        handle for ``finish_proposals``; the caller may launch the next video's forward first, so that the host-side wait and
        the launch latency of the next forward overlap with GPU work (``run`` does)."""
        logits, offsets, masks = flat
        t0 = time.perf_counter()
        # the pyramid (and its points) starts at T / vid_stride (video_net.py:59-74); segments go back to input clips in finish_proposals
        assert window_ext is None or self.vid_stride == 1, 'ext_scores are per input clip: the reference multiplies them at level 0 (worker_v2.py:1150-1156)'
        segs, scores, counts = _nms.collect_segments(logits, offsets, masks, T // self.vid_stride, self.num_fpn_levels, self.pre_nms_thresh,
                                                     self.pre_nms_topk, self.seg_len_thresh, ext_scores=window_ext)
        self.time_dict['post_process'].append(time.perf_counter() - t0)
        cfg = self.nms_cfg
        if cfg.get('max_num_segs', 0) <= 0:             # keeps nothing in the reference either (nms.py:144-146): per query on the host
            return ('eager', segs, scores, counts, data)
        t0 = time.perf_counter()
        # every query of the video in one pass on the device
        s_all, c_all, k_all = _nms.batched_nms_queries(segs, scores, counts, **cfg)
        nq, M = c_all.shape
        packed = torch.cat((s_all.reshape(nq, 2 * M), c_all, k_all[:, None].to(c_all.dtype)), 1)
        # the pinned buffer belongs to the handle until finish_proposals has read it, then goes back to the free list of its
        # shape: two handles never share one, whatever order shapes arrive in (nq differs between videos)
        free = self._pinned.setdefault(tuple(packed.shape), [])
        host = free.pop() if free else torch.empty(packed.shape, dtype=packed.dtype, pin_memory=True)
        host.copy_(packed, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        self.time_dict['nms'].append(time.perf_counter() - t0)
        return ('packed', done, host, nq, M, data)

    def finish_proposals(self, handle):
        """Wait for a video's kept rows and turn them into the reference's result list (worker_v2.py:1120-1124)."""
        if handle[0] == 'eager':
            _, segs, scores, counts, data = handle
            counts_h = counts.cpu()
            results = []
            for q in range(segs.shape[0]):
                n = int(counts_h[q])
                s, c = _nms.batched_nms(segs[q, :n], scores[q, :n], **self.nms_cfg)
                results.append({'segments': s.cpu(), 'scores': c.cpu()})
            return results
        _, done, host, nq, M, data = handle
        done.synchronize()                              # the only wait
        packed = host.clone()
        self._pinned.setdefault(tuple(host.shape), []).append(host)      # back to the free list: a later video may take it now
        s_host = packed[:, :2 * M].reshape(nq, M, 2)
        if data is not None:                            # on the <= max_num_segs kept rows, on the host: the same fp32 operations
            s_host = s_host * self.vid_stride
            s_host = (s_host * data['clip_stride'] + 0.5 * data['clip_size']) / data['fps']        # worker_v2.py:1120-1122
            s_host = torch.clamp(s_host, min=0, max=data['duration'])
        results = []
        for q in range(nq):
            k = int(packed[q, 3 * M])
            results.append({'segments': s_host[q, :k], 'scores': packed[q, 2 * M:3 * M][:k]})
        return results

    def generate_proposals(self, flat, T, data=None, window_ext=None):
        """_collect_segments + batched_nms + seconds for every query; results as in worker_v2.py:1124.
        ``window_ext``: padded external scores (NQ, T) on the device (worker_v2.py:1078-1081) or None."""
        return self.finish_proposals(self.launch_proposals(flat, T, data, window_ext))

    def _own_carry(self, on=True):
        """while predict / run are in charge, the model does not raise for the one-pass LayerNorm guard at its next call: the evaluator
        repeats the affected videos (replicas made meanwhile inherit the setting)"""
        self.model._carry_handled_by_caller = bool(on)

    def predict(self, data):
        self._own_carry()
        try:
            return self._predict(data)
        finally:
            self._own_carry(False)

    def _predict(self, data):
        flat, T = self.forward(data)
        res = self.generate_proposals(flat, T, data, self._window_ext)
        if self._carry_tripped([self.model]):       # the host has just synchronised for the proposals: 4 more bytes
            flat, T = self.forward(data)            # one-pass LayerNorm guard: the model runs two-pass launches now, the video is repeated
            res = self.generate_proposals(flat, T, data, self._window_ext)
            self._check_numerics([self.model])
        return res

    def run(self, dataset, counter: RecallCounter = None, n_streams: int = 1, batch_videos: int = 1):
        """Evaluator.run (worker_v2.py:815-910) over an iterable of per-video dicts (keys as in
        libs/data/dataset.py:977-994: vid, shallow_vid, text, text_cls, segment, fps, clip_stride, clip_size, duration).
        ``batch_videos > 1``: consecutive videos of the same padded length (every video up to max_vid_len is one) share a
        forward (``model.forward_videos``); same proposals, fewer and larger kernel launches."""
        self._own_carry()
        try:
            return self._run(dataset, counter, n_streams, batch_videos)
        finally:
            self._own_carry(False)

    def _run(self, dataset, counter, n_streams, batch_videos):
        counter = counter or RecallCounter(self.opt['eval'].get('ranks', (1, 5)), self.opt['eval'].get('iou_threshs', (0.3, 0.5)))
        if batch_videos > 1:
            pending = []                              # (data, args, T, ext)

            def flush():
                if not pending:
                    return
                T = pending[0][2]
                t0 = time.perf_counter()

                def compute():
                    self.model.forward_videos([p[1] for p in pending])
                    logits, offsets, masks = self.model._last_flat
                    q, out = 0, []
                    for data, args, _, ext in pending:
                        n = len(args[3])
                        out.append(self.generate_proposals((logits[q:q + n], offsets[q:q + n], masks[q:q + n]), T, data, ext))
                        q += n
                    return out

                results = compute()
                if self._carry_tripped([self.model]):          # (the host has synchronised for the proposals) repeat the group
                    results = compute()
                self.time_dict['forward'].append(time.perf_counter() - t0)
                for res, (data, _, _, _) in zip(results, pending):
                    counter.update(res, data['segment'])
                pending.clear()

            for data in dataset:
                args, T, ext = self.prepare(data)
                if pending and (pending[0][2] != T or len(pending) == batch_videos):
                    flush()
                pending.append((data, args, T, ext))
            flush()
            self._check_numerics([self.model])
            return counter
        if n_streams <= 1:
            # one video per forward like the reference, software-pipelined by one video: the proposals of video i are waited for
            # after the forward of video i + 1 has been launched (same stream: its kernels run after video i's decode / NMS)
            def launch(data):
                flat, T = self.forward(data)
                return self.launch_proposals(flat, T, data, self._window_ext)

            prev = None
            for data in dataset:
                handle = launch(data)
                if prev is not None:
                    res = self.finish_proposals(prev[0])
                    # the one-pass LayerNorm guard, without a wait: the probe of prev's forward has landed (its proposals have); if it
                    # -- or the forward in flight behind it -- tripped, the model switches to two-pass launches and both are repeated
                    if self.model.ln_carry_flag_nowait():
                        torch.cuda.synchronize()
                        self.finish_proposals(handle)          # (gives its pinned buffer back)
                        self._carry_tripped([self.model])
                        res = self.finish_proposals(launch(prev[1]))
                        handle = launch(data)
                    counter.update(res, prev[1]['segment'])
                prev = (handle, data)
            if prev is not None:
                res = self.finish_proposals(prev[0])
                if self._carry_tripped([self.model]):
                    res = self.finish_proposals(launch(prev[1]))
                counter.update(res, prev[1]['segment'])
            self._check_numerics([self.model])
            return counter
        # throughput mode: n_streams videos in flight, one model replica (shared parameters, own workspace) and one HIP
        # stream each; the decode / NMS of a video (host syncs on its own stream only) overlaps the forwards of the others
        lanes = [(self.model if i == 0 else self.model.replica(), torch.cuda.Stream()) for i in range(n_streams)]
        group = []

        def finish():
            results = []
            for stream, mdl, flat, T, data, ext in group:
                with torch.cuda.stream(stream):
                    results.append(self.generate_proposals(flat, T, data, ext))
            if self._carry_tripped([m for m, _ in lanes]):     # (every lane's host wait is behind us) repeat the group, two-pass LayerNorms
                results = []
                for stream, mdl, _, _, data, _ in group:
                    with torch.cuda.stream(stream):
                        flat, T = self.forward(data, mdl)
                        results.append(self.generate_proposals(flat, T, data, self._window_ext))
            for res, entry in zip(results, group):
                counter.update(res, entry[4]['segment'])
            group.clear()

        for data in dataset:
            mdl, stream = lanes[len(group)]
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                flat, T = self.forward(data, mdl)
            group.append((stream, mdl, flat, T, data, self._window_ext))
            if len(group) == n_streams:
                finish()
        finish()
        torch.cuda.synchronize()
        self._check_numerics([m for m, _ in lanes])
        return counter

    @staticmethod
    def _carry_tripped(models):
        """The one-pass LayerNorm guard (dcf_numerics_status & 16: a carried LayerNorm met a row whose mean dwarfs its spread; that forward's
        logits were set to NaN) checked where the host has synchronised anyway.  True: EVERY model now runs the two-pass LayerNorm launches
        (set_ln_carry(False)), the flags are reset, and the caller repeats the forwards since the last check -- the run goes on instead of
        aborting.  The fp16-range bit still raises (nothing to fall back to but another gemm_mode)."""
        tripped = False
        for m in models:
            st = m.numerics_status(reset=False) if hasattr(m, 'numerics_status') else 0
            if st & 1:
                m.numerics_status(reset=True)
                raise RuntimeError("an activation left the fp16 operand range of the f16x3 GEMM mode: results are not valid; "
                                   "set opt.model.gemm_mode = 'bf16x6'")
            tripped = tripped or bool(st & 16)
        if tripped:
            for m in models:
                m.acknowledge_ln_carry()
        return tripped

    @staticmethod
    def _check_numerics(models):
        """the f16x3 GEMM mode flags operands beyond the fp16 range (|a| >= 4094) instead of passing inf / NaN on"""
        for m in models:
            st = m.numerics_status(reset=True) if hasattr(m, 'numerics_status') else 0
            if st & 1:
                raise RuntimeError("an activation left the fp16 operand range of the f16x3 GEMM mode: results are not valid; "
                                   "set opt.model.gemm_mode = 'bf16x6'")
            if st & 16:
                for mm in models:
                    mm.set_ln_carry(False)
                raise RuntimeError("a LayerNorm carried between kernels as one-pass row statistics met a row whose mean dwarfs its spread: "
                                   "results are not valid; the model now runs two-pass LayerNorm launches (set_ln_carry(False)) -- run again")
