"""Multi-GPU execution of the grounding path.  One process per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

Two granularities:

1. **Independent units** -- every (video, query) pair is independent (libs/modeling/model.py:526;
   videos are batch_size 1, libs/worker_v2.py:739).  ``assign_units`` bin-packs videos by length
   onto ranks; no data-path collective is needed (bench.py --gpus N runs this way).

2. **One long video sharded over T** -- ``sharded_forward``.  Rank r owns the clips
   ``[lo_r, hi_r)`` and computes the window ``[lo_r - H, hi_r + H)`` (overlap-recompute, H >=
   the one-sided receptive field of the network, rounded up to the pyramid alignment so that
   stride-2 phases and attention windows coincide with the unsharded run).  Exactly two
   collectives:
     AG-1  all-gather of the raw sidekick scores (NQ x T/W fp32 per rank): the block top-k gate
           (model.py:531-541) is the only global reduction of the network; every rank then runs
           the same deterministic selection on identical data.
     AG-2  all-gather of the owned slice of every pyramid level's logits / offsets / masks.
   Everything else (per-position LayerNorm, 1x1 convs, cross-attention against the replicated
   text, k3 convs, window attention, stride-2 pooling, the TCN) is position-local, so the owned
   outputs equal the unsharded ones up to fp32 round-off of identical per-position operations.

The compute is injected through a small backend object so the same orchestration runs on the
HIP model (``HipBackend``) and, in the CPU tests, on any callable with the same contract.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch


# ---------------------------------------------------------------------------------------------
# independent units
# ---------------------------------------------------------------------------------------------
def assign_units(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Longest-processing-time bin packing of videos (cost ~ T * NQ) onto ``world`` ranks.
    Returns, per rank, the list of unit indices; deterministic."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    load = [0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += lengths[i]
    return out


# ---------------------------------------------------------------------------------------------
# T-sharding
# ---------------------------------------------------------------------------------------------
def alignment(n_levels: int, win: int) -> int:
    """Shard boundaries must keep every level's stride-2 phase and window chunking:
    multiples of 2^(L-1) * max(win // 2, 1) (cf. libs/worker_v2.py:769-781)."""
    return (2 ** (n_levels - 1)) * max(win // 2, 1)


def receptive_field(n_levels: int, win: int, fusion_layers: int = 2, n_embd_convs: int = 2, n_stem: int = 0,
                    head_layers: int = 2) -> int:
    """Conservative one-sided receptive field of a level-0 output, in level-0 clips."""
    L, hw = n_levels, win // 2
    r = fusion_layers + n_embd_convs                       # depthwise k3 per fusion layer, dense k3 embedding convs
    r += n_stem * (1 + hw)
    for l in range(L):
        s = 2 ** l
        r += s * hw                                        # window attention at level l
        r += s if l == 0 else s // 2 + s                   # depthwise k3 (input-level units) + max-pool skip
    r += (head_layers + 1) * 2 ** (L - 1)                  # cls_head trunk + output conv at the coarsest level
    r += 2 ** (L - 1) + (2 ** L - 1)                       # nearest upsampling granularity + dilated TCN
    r += 2 ** (L - 1)                                      # pooling the refined logits down the pyramid
    r += (head_layers + 1) * 2 ** (L - 1)                  # cls_head2 / reg_head
    return r


def shard_plan(T: int, world: int, n_levels: int, win: int, halo: int) -> List[Tuple[int, int, int, int]]:
    """(lo, hi, w_lo, w_hi) per rank: owned range and computed window, all multiples of the alignment."""
    a = alignment(n_levels, win)
    assert T % a == 0, f'T={T} must be a multiple of {a}'
    units = T // a
    halo = -(-halo // a) * a
    plan = []
    for r in range(world):
        lo = (units * r // world) * a
        hi = (units * (r + 1) // world) * a
        plan.append((lo, hi, max(0, lo - halo), min(T, hi + halo)))
    return plan


class HipBackend:
    """Compute steps of ``sharded_forward`` on the MI355X model (cvpr2025-decafnet_amd.modeling)."""

    def __init__(self, model):
        self.model = model

    def scores(self, shallow_own, text_cls):
        from . import _lib
        lib = _lib.lib()
        D, T = shallow_own.shape
        nq = text_cls.shape[0]
        sh = shallow_own.contiguous()
        out = torch.empty(nq, T, device=sh.device, dtype=torch.float32)
        _lib.check(lib.dcf_op_sidekick(_lib.ptr(sh), _lib.ptr(text_cls.contiguous()), _lib.ptr(out), D, T, nq,
                                       int(self.model.norm), _lib.current_stream()), 'dcf_op_sidekick')
        return out

    def gate(self, correl_full, mask_full):
        from . import _lib
        lib = _lib.lib()
        nq, T = correl_full.shape
        gate = torch.empty(nq, T, device=correl_full.device, dtype=torch.float32)
        mo = torch.empty(nq, T, device=correl_full.device, dtype=torch.bool)
        _lib.check(lib.dcf_op_gate(_lib.ptr(correl_full.contiguous()), _lib.ptr(mask_full.contiguous()), _lib.ptr(gate),
                                   _lib.ptr(mo), T, nq, self.model.sn, float(self.model.sratio), 1, _lib.current_stream()),
                   'dcf_op_gate')
        return gate

    def forward_window(self, vid_w, shallow_w, mask_w, texts, tmasks, gate_w, T_global, w_lo):
        pe = None
        if self.model.vid_net.use_abs_pe:
            pe = self.model.full_position_encoding(T_global, vid_w.device)[w_lo:w_lo + vid_w.shape[-1]]
        self.model.forward_window(vid_w[None], shallow_w[None], mask_w[None], texts, tmasks, gate_w, pe)
        return self.model._last_flat          # (nq, S_w), (nq, S_w, 2), (nq, S_w)


def _all_gather_cat(x: torch.Tensor, dim: int, group=None) -> torch.Tensor:
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = [torch.zeros(1, dtype=torch.int64, device=x.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([x.shape[dim]], dtype=torch.int64, device=x.device), group=group)
    sizes = [int(s) for s in sizes]
    mx = max(sizes)
    pad_shape = list(x.shape)
    pad_shape[dim] = mx
    buf = x.new_zeros(pad_shape)
    buf.narrow(dim, 0, x.shape[dim]).copy_(x)
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf.contiguous(), group=group)
    return torch.cat([o.narrow(dim, 0, s) for o, s in zip(outs, sizes)], dim=dim)


def sharded_forward(backend, vid_w, shallow_w, mask_full, plan_r, T, n_levels, texts, text_cls, tmasks, group=None):
    """Eval forward of ONE video sharded over the ranks of ``group``.

    vid_w, shallow_w : (D, w_hi - w_lo) this rank's window of the (zero padded) features
    mask_full        : (T,) bool validity of every clip of the whole video (cheap, replicated)
    plan_r           : this rank's (lo, hi, w_lo, w_hi) from ``shard_plan``
    Returns the whole video's outputs on every rank, exactly like ``model(..., eval=True)``:
    three lists (NQ) of L-tuples  logits (1, T_l), offsets (1, T_l, 2), masks (1, T_l).
    """
    lo, hi, w_lo, w_hi = plan_r
    nq = text_cls.shape[0]
    # AG-1: raw sidekick scores of the owned clips -> whole video on every rank
    own = shallow_w[:, lo - w_lo:hi - w_lo]
    correl = _all_gather_cat(backend.scores(own, text_cls), dim=1, group=group)            # (nq, T)
    assert correl.shape == (nq, T)
    gate_full = backend.gate(correl, mask_full)                                            # identical on every rank
    gate_w = gate_full[:, w_lo:w_hi].contiguous()
    logits, offsets, masks = backend.forward_window(vid_w, shallow_w, mask_full[w_lo:w_hi].contiguous(), texts, tmasks,
                                                    gate_w, T, w_lo)
    # AG-2: owned slice of every level
    Tw = w_hi - w_lo
    out_l, out_o, out_m = [], [], []
    off = 0
    for l in range(n_levels):
        Tl = Tw >> l
        a, b = (lo - w_lo) >> l, (hi - w_lo) >> l
        out_l.append(_all_gather_cat(logits[:, off + a:off + b].contiguous(), 1, group))
        out_o.append(_all_gather_cat(offsets[:, off + a:off + b].contiguous(), 1, group))
        out_m.append(_all_gather_cat(masks[:, off + a:off + b].to(torch.uint8).contiguous(), 1, group).bool())
        off += Tl
    lg = [tuple(x[q][None] for x in out_l) for q in range(nq)]
    of = [tuple(x[q][None] for x in out_o) for q in range(nq)]
    mk = [tuple(x[q][None] for x in out_m) for q in range(nq)]
    return lg, of, mk
