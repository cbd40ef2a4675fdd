"""Deterministic synthetic weights and inputs (there are no checkpoints or datasets here).

The reference initialises every ``LayerScale`` to 1e-4 (libs/modeling/blocks.py:675-678),
which scales each residual branch to ~1e-4, so a default-initialised model barely
exercises the attention/FFN numerics.  ``make_state_dict`` instead fills a state_dict with
O(1) values from a seeded generator, keyed by parameter NAME and SHAPE only, so the exact
same tensors can be regenerated on any machine from ``(shapes, seed)``.
"""
from __future__ import annotations

import math
from typing import Dict, Sequence

import torch


def _fill(name: str, shape: Sequence[int], g: torch.Generator) -> torch.Tensor:
    shape = tuple(shape)
    r = torch.randn(shape, generator=g, dtype=torch.float32) if len(shape) else \
        torch.randn((), generator=g, dtype=torch.float32)
    leaf = name.rsplit('.', 1)[-1]
    if leaf == 'scale':                      # LayerScale (1,C,1) or reg-head Scale ()
        return 1.0 + 0.25 * r
    if leaf == 'bkgd_token':
        return 0.5 * r
    if leaf == 'weight':
        if len(shape) == 3:                  # conv (out, in/groups, k)
            fan_in = shape[1] * shape[2]
            return r / math.sqrt(fan_in)
        return 1.0 + 0.1 * r                 # LayerNorm (C,1) / nn.LayerNorm (C,)
    if leaf == 'bias':
        return 0.1 * r
    raise KeyError(f'unknown parameter kind: {name} {shape}')


def make_state_dict(shapes: Dict[str, Sequence[int]], seed: int) -> Dict[str, torch.Tensor]:
    """Seeded O(1) weights; iteration order is the sorted key order (not dict order)."""
    g = torch.Generator().manual_seed(int(seed))
    return {k: _fill(k, shapes[k], g) for k in sorted(shapes)}


def make_inputs(D: int, T: int, vid_len: int, nq: int, text_in: int, lq: int, seed: int):
    """vid/shallow ~ N(0,1) (1,D,T) with the padded tail zeroed, text_cls (NQ,D), raw text
    tokens NQ x (C_t, Lq).  SURVEY.md 8(d)."""
    g = torch.Generator().manual_seed(int(seed))
    vid = torch.randn(1, D, T, generator=g)
    shallow = torch.randn(1, D, T, generator=g)
    vid[..., vid_len:] = 0
    shallow[..., vid_len:] = 0
    mask = (torch.arange(T) < vid_len).view(1, T)
    text_cls = torch.randn(nq, D, generator=g)
    tokens = [torch.randn(text_in, lq, generator=g) for _ in range(nq)]
    return dict(vid=vid, shallow_vid=shallow, vid_masks=mask, text_cls=text_cls, tokens=tokens)
