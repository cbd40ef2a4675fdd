"""Real-data side of the evaluation harness: the feature-file formats, the text-CLS table, the
annotation file and the per-video dict of the reference's video-centric dataset, as far as ``Evaluator.run`` consumes them
(libs/data/dataset.py).  Pure host code (numpy / torch CPU tensors): the tensors it yields are what
``GroundingEvaluator.predict`` / ``run`` take.

    clip features      VID_LOAD_FUNC, dataset.py:107-135: '<id>.npy' (T, C), '<id>.pt' (T, C) tensor, '<id>.pk' = pickled
                       sequence of (T, C) arrays ('pk0' / 'pk1' pick an entry, 'pk_avg' averages the first two)
    video features     _load_vid_feats, dataset.py:362-408: several feature sources of one video aligned (the shorter ones
                       padded by repeating their last clip, lengths may differ by <= 10), concatenated along channels,
                       temporally down-sampled, returned channel-major (C, T), optionally L2-normalised per clip
    text features      _load_text_feats, dataset.py:460-484: '<text_id>.npy' (L, C) -> (C, L)
    text-CLS table     dataset.py:556-560, 671-678: np.save'd dict sentence -> (1, D) array, one file per split
    annotations        _parse_annotations, dataset.py:287-353: {split: {video: {fps, num_frames, [duration], annotations:
                       [{segment, sentence, [sentence_id]}]}}}; segments clipped to [0, duration], empty ones dropped
    per-video dict     __getitem__, dataset.py:977-994 (eval: every query of a video in one sample, :599-603)
"""
from __future__ import annotations

import json
import os
import pickle
from collections import OrderedDict
from typing import Dict, Iterator, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

FEATURE_FORMATS = ('npy', 'pt', 'pk0', 'pk1', 'pk_avg')


def load_clip_features(path_without_ext: str, fmt: str = 'npy') -> np.ndarray:
    """One feature source of one video as stored on disk: (T, C)."""
    if fmt == 'npy':
        return np.load(path_without_ext + '.npy').astype(np.float32)
    if fmt == 'pt':
        return torch.load(path_without_ext + '.pt').numpy()
    if fmt in ('pk0', 'pk1', 'pk_avg'):
        with open(path_without_ext + '.pk', 'rb') as fh:
            entries = pickle.load(fh)
        if fmt == 'pk_avg':
            return (entries[0] + entries[1]) / 2
        return entries[int(fmt[2])]
    raise NotImplementedError(f'feature format {fmt!r}: one of {FEATURE_FORMATS}')


def load_video_features(feat_dirs: Sequence[str], vid_id: str, fmt: str = 'npy', downsample_rate: int = 1,
                        normalize: bool = False) -> torch.Tensor:
    """All feature sources of a video -> (C_total, T) float tensor, the reference's in-memory layout."""
    srcs = [load_clip_features(os.path.join(d, vid_id), fmt) for d in feat_dirs]
    if len(srcs) > 1:
        longest = max(len(a) for a in srcs)
        if longest - min(len(a) for a in srcs) > 10:
            raise ValueError(f'misaligned features ([max] {longest}, [min] {min(len(a) for a in srcs)}) for video {vid_id}')
        srcs = [a if len(a) == longest else np.concatenate((a, np.tile(a[-1], (longest - len(a), 1)))) for a in srcs]
        feats = np.concatenate(srcs, axis=-1)
    else:
        feats = srcs[0]
    if downsample_rate > 1:
        feats = feats[::downsample_rate]
    out = torch.from_numpy(np.ascontiguousarray(feats.transpose()))
    return F.normalize(out, dim=0) if normalize else out


def load_text_features(text_feat_dir: str, text_id, normalize: bool = False) -> torch.Tensor:
    """Token features of one sentence: (C, L)."""
    a = np.load(os.path.join(text_feat_dir, str(text_id) + '.npy')).astype(np.float32)
    out = torch.from_numpy(np.ascontiguousarray(a.transpose()))
    return F.normalize(out, dim=0) if normalize else out


class TextClsTable:
    """sentence -> sentence-level (CLS) feature, merged over the files of all splits."""

    def __init__(self, fnames: Sequence[str]):
        self.table: Dict[str, np.ndarray] = {}
        for f in fnames:
            self.table.update(np.load(f, allow_pickle=True).item())

    def lookup(self, sentences: Sequence[str]) -> torch.Tensor:
        """(n, D): the rows of the sentences, concatenated like dataset.py:675-677"""
        return torch.from_numpy(np.concatenate([self.table[s] for s in sentences], axis=0))


def parse_annotations(anno_file: str, splits: Sequence[str], downsample_rate: int = 1):
    """-> OrderedDict video -> {fps, num_frames, num_clips, duration, text_ids, segments (n, 2), annotations}."""
    with open(anno_file) as fh:
        anno = json.load(fh)
    merged = {}
    for s in splits:
        if s not in anno:
            raise KeyError(f'split [{s}] does not exist')
        merged.update(anno[s])
    videos = OrderedDict()
    for key, value in merged.items():
        if 'annotations' not in value:
            continue
        fps, num_frames = float(value['fps']), int(value['num_frames'])
        duration = float(value['duration']) if 'duration' in value else num_frames / fps
        num_clips = (value['num_clips'] + downsample_rate - 1) // downsample_rate if 'num_clips' in value else None
        text_ids, segments = [], []
        for i, pair in enumerate(value['annotations']):
            start, end = max(float(pair['segment'][0]), 0), min(float(pair['segment'][1]), duration)
            if end - start <= 0:
                continue
            text_ids.append(pair.get('sentence_id', key + '_{:04d}'.format(i)))
            segments.append((start, end))
        if not text_ids:
            continue
        videos[key] = dict(fps=fps, num_frames=num_frames, num_clips=num_clips, duration=duration, text_ids=tuple(text_ids),
                           segments=np.array(segments), annotations=value['annotations'])
    return videos


class VideoCentricEvalData:
    """Iterable over the evaluation samples of the reference's ``VideoCentricTwoFeatDataset`` (one sample = one video with
    all of its queries), yielding the dict ``Evaluator.simple_predict`` / ``GroundingEvaluator.predict`` consume.

    ``cfg`` carries the ``opt.data`` keys the reference reads: anno_file, eval_split (or split), vid_feat_dir,
    shallow_vid_feat_dir, text_feat_dir, text_cls_fname (with ``{split}``), vid_load, shallow_vid_load, downsample_rate,
    clip_size, clip_stride, normalize_vid, normalize_text, ext_score_dir / normalize_scores / temperature (optional)."""

    def __init__(self, cfg):
        g = cfg.get
        splits = g('eval_split', g('split', ('val',)))
        self.splits = (splits,) if isinstance(splits, str) else tuple(splits)
        self.ds = int(g('downsample_rate', 1))
        self.videos = parse_annotations(cfg['anno_file'], self.splits, self.ds)
        self.vid_dirs, self.shallow_dirs = list(cfg['vid_feat_dir']), list(cfg['shallow_vid_feat_dir'])
        self.vid_fmt, self.shallow_fmt = g('vid_load', 'npy'), g('shallow_vid_load', 'npy')
        self.text_dir = cfg['text_feat_dir']
        self.clip_size = g('clip_size', 16)
        self.clip_stride = g('clip_stride', 16) * self.ds                     # dataset.py:240
        self.normalize_vid, self.normalize_text = bool(g('normalize_vid', False)), bool(g('normalize_text', False))
        self.ext_score_dir = g('ext_score_dir')
        self.normalize_scores, self.temperature = bool(g('normalize_scores', False)), float(g('temperature', 1.0))
        self.cls = TextClsTable([cfg['text_cls_fname'].format(split=s) for s in self.splits])

    def __len__(self):
        return len(self.videos)

    def sample(self, vid_id: str):
        v = self.videos[vid_id]
        n = len(v['segments'])
        vid = load_video_features(self.vid_dirs, vid_id, self.vid_fmt, self.ds, self.normalize_vid)
        # the sidekick features are stored at the down-sampled rate already (dataset.py:1033,1066-1068)
        shallow = load_video_features(self.shallow_dirs, vid_id, self.shallow_fmt, 1, self.normalize_vid)
        text = tuple(load_text_features(self.text_dir, t, self.normalize_text) for t in v['text_ids'])
        # the reference indexes the raw annotation list with the index of the kept segment (dataset.py:672-674)
        text_cls = self.cls.lookup([v['annotations'][i]['sentence'] for i in range(n)])
        ext: Optional[torch.Tensor] = None
        if self.ext_score_dir is not None:                                   # _load_ext_scores, dataset.py:486-505
            rows = []
            for t in v['text_ids']:
                sc = np.load(os.path.join(self.ext_score_dir, str(t) + '.npy')).astype(np.float32)[::self.ds]
                sc = torch.from_numpy(np.ascontiguousarray(sc))[None]
                rows.append(torch.sigmoid(sc / self.temperature) if self.normalize_scores else sc)
            ext = torch.cat(rows)
        return dict(fps=v['fps'], num_frames=v['num_frames'], duration=v['duration'], segment=v['segments'],
                    clip_size=self.clip_size, clip_stride=self.clip_stride, clip_id=vid_id, text_id=tuple(range(n)),
                    vid=vid, shallow_vid=shallow, text=text, text_cls=text_cls, ext_scores=ext)

    def __iter__(self) -> Iterator[dict]:
        for vid_id in self.videos:
            yield self.sample(vid_id)
