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
   collectives, each ONE all-gather whose piece sizes every rank derives from the shard plan:
     AG-1  all-gather of the raw sidekick scores (NQ x T/W fp32 per rank): the block top-k gate
           (model.py:531-541) is the only global reduction of the network; every rank then runs
           the same deterministic selection on identical data.
     AG-2  all-gather of the owned slice of every pyramid level, packed (logit, offset0, offset1, mask).
   Everything else (per-position LayerNorm, 1x1 convs, cross-attention against the replicated
   text, k3 convs, window attention, stride-2 pooling, the TCN) is position-local, so the owned
   outputs equal the unsharded ones up to fp32 round-off of identical per-position operations.

3. **Both at once** -- ``shard_plan_2d`` / ``sharded_forward_2d``: the queries of the video are dealt to query groups first
   (no recompute at all along that axis), only the remaining factor of the world size cuts the clip axis; one more
   static-size all-gather collects the other groups' outputs.

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


# the layer counts of BASELINE.md's probe configuration (and of every test fixture that does not say otherwise): for callers that plan
# for THAT architecture without a model at hand; with a model, pass **arch_of(model)
PROBE_ARCH = dict(fusion_layers=2, n_embd_convs=2, n_stem=0, head_layers=2)


def receptive_field(n_levels: int, win: int, *, fusion_layers: int, n_embd_convs: int, n_stem: int, head_layers: int) -> int:
    """(The layer counts have no defaults, like ``hybrid_halos``: a halo computed for another architecture is silently wrong.)
    Exact one-sided receptive field of an output position, in level-0 clips (the larger, LEFT reach; the right reach
    is 2^(L-1) smaller because nearest upsampling reads the sample at or left of a clip).  Validated against a
    perturbation probe (tools/receptive_field.py); 2304 for L = 8, w = 9 (right reach 2176).

    Encoder pyramid: fusion / embedding k3 convolutions reach 1 each; level 0 adds its depthwise k3 (1) and window
    attention (w//2); level l >= 1 works on a grid of 2^l clips: stride-2 depthwise k3 / max-pool reach one input sample
    (2^(l-1)), attention w//2 samples (2^l * w//2).  Heads: (head_layers + 1) k3 convolutions on the level's grid.
    Refinement (model.py:442-471) at the coarsest grid W = 2^(L-1): cls_head (hl+1)W, nearest upsampling + dilated TCN
    (2W - 1) + pooling back down (W - 1) land on a multiple of W after flooring: 3W, then cls_head2 / reg_head (hl+1)W."""
    L, hw = n_levels, win // 2
    r = fusion_layers + n_embd_convs + n_stem * (1 + hw)
    r += 1 + hw
    for l in range(1, L):
        r += 2 ** (l - 1) + hw * 2 ** l
    W = 2 ** (L - 1)
    return (2 * head_layers + 5) * W + r


def shard_plan(T: int, world: int, n_levels: int, win: int, halo: int) -> List[Tuple[int, int, int, int]]:
    """(lo, hi, w_lo, w_hi) per rank: owned range and computed window.  Owned ranges are multiples of the alignment;
    a window starts / ends on the stride-2 phase grid (multiples of 2^(L-1)) at least `halo` clips beyond the owned range
    and is lengthened to a multiple of the alignment (blocks.py:216: T_l % (w//2) == 0 at every level)."""
    a = alignment(n_levels, win)
    ph = 2 ** (n_levels - 1)
    assert T % a == 0, f'T={T} must be a multiple of {a}'
    units = T // a
    halo = -(-halo // ph) * ph
    plan = []
    for r in range(world):
        lo = (units * r // world) * a
        hi = (units * (r + 1) // world) * a
        w_lo, w_hi = max(0, lo - halo), min(T, hi + halo)
        extra = (-(w_hi - w_lo)) % a
        grow = min(extra, T - w_hi)                      # lengthen to the right first, then to the left
        w_hi += grow
        w_lo -= extra - grow
        assert w_lo >= 0 and (w_hi - w_lo) % a == 0 and w_lo % ph == 0
        plan.append((lo, hi, w_lo, w_hi))
    return plan


def shard_plan_2d(T: int, world: int, nq: int, n_levels: int, win: int, halo: int, hybrid_arch: dict = None):
    """Queries first, clips second.  The (video, query) pairs of ONE video are independent after the (query-independent)
    vid_map products, so the ranks form a grid of ``q_groups`` query groups x ``t_shards`` clip chunks: rank r = qg * t_shards + t
    computes the window of clip chunk t for the queries of group qg.  Overlap-recompute costs rows only along the clip axis:
    with NQ = 8 queries on 8 ranks nothing is cut at all (1.00x the rows of an even share), NQ = 4 cuts T in two (1.16x at
    T = 65 536), NQ = 1 is the pure T-shard (1.625x).  Returns the dict of the plan with the smallest number of rows per rank
    (ties: fewer clip chunks):
      t_shards, q_groups, plan (``shard_plan`` of the clip axis), queries (per group: (q_lo, q_hi)), rows_factor.
    ``hybrid_arch`` (dict of ``hybrid_halos``' layer counts, all four of them: ``arch_of(model)``): the clip axis may also be cut with the pyramid split at a level k
    (``hybrid_plan``: no recomputed pyramid top, 1.12x instead of 1.56x at T = 65 536 on 8 clip chunks); the candidate then carries
    ``hybrid`` = that plan, its ``plan`` entries are (lo, hi, n_lo, n_hi) -- the NARROW window, what a caller slices its features to --
    and ``sharded_forward_2d`` runs ``hybrid_forward`` inside a clip-chunk group.
    Everything is a function of (T, world, nq, L, w): nothing is negotiated at run time."""
    best = None
    for ts in range(1, world + 1):
        if world % ts:
            continue
        qs = world // ts
        if qs > nq:
            continue
        plan = shard_plan(T, ts, n_levels, win, halo) if ts > 1 else [(0, T, 0, T)]
        per = [(nq * g // qs, nq * (g + 1) // qs) for g in range(qs)]
        rows = max(p[3] - p[2] for p in plan) * max(b - a for a, b in per)
        hyb = None
        if hybrid_arch is not None and ts > 1 and n_levels > 2:
            hp = hybrid_plan(T, ts, n_levels, win, None, **hybrid_arch)
            hrows = hp['rows_factor'] * (T / ts) * max(b - a for a, b in per)       # in level-0 clips of an even share, like `rows`
            if hrows < rows:
                hyb, rows = hp, hrows
                plan = [(r['lo'], r['hi'], r['n_lo'], r['n_hi']) for r in hp['ranks']]
        cand = dict(t_shards=ts, q_groups=qs, plan=plan, queries=per, rows_factor=rows / (T * nq / world), hybrid=hyb)
        if best is None or rows < best[0] or (rows == best[0] and ts < best[1]['t_shards']):
            best = (rows, cand)
    assert best is not None, f'no (query, clip) grid for world={world}, nq={nq}'
    return best[1]


def make_grid_groups(t_shards: int, q_groups: int):
    """The process groups of ``sharded_forward_2d``; every rank of the world calls this with the same arguments (groups are
    created collectively).  Returns (t_groups, q_groups_): t_groups[qg] = the ranks that share query group qg (one clip chunk
    each), q_groups_[t] = the ranks that share clip chunk t (one query group each)."""
    import torch.distributed as dist
    tg = [dist.new_group([g * t_shards + t for t in range(t_shards)]) for g in range(q_groups)]
    qg = [dist.new_group([g * t_shards + t for g in range(q_groups)]) for t in range(t_shards)]
    return tg, qg


class HipBackend:
    """Compute steps of ``sharded_forward`` on the MI355X model (cvpr2025-decafnet_amd.modeling)."""

    def __init__(self, model):
        self.model = model

    def arch(self):
        return arch_of(self.model)

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


    # the three phases of ``hybrid_forward`` (dcf_hybrid_phase1 / 2 / 3 through the model)
    def hybrid_phase1(self, vid_w, shallow_w, mask_w, texts, tmasks, gate_w, T_global, n_lo, k, Tc):
        pe = None
        if self.model.vid_net.use_abs_pe:
            pe = self.model.full_position_encoding(T_global, vid_w.device)[n_lo:n_lo + vid_w.shape[-1]]
        return self.model.hybrid_phase1(vid_w, shallow_w, mask_w, texts, tmasks, gate_w, k, Tc, pe)

    def hybrid_phase2(self, featk_c, maskk_c, off_k):
        return self.model.hybrid_phase2(featk_c, maskk_c, off_k)

    def hybrid_phase3(self, refk_c):
        return self.model.hybrid_phase3(refk_c)


def _gather_static(x: torch.Tensor, sizes: Sequence[int], group=None) -> List[torch.Tensor]:
    """ONE all-gather of per-rank pieces whose lengths along the LAST-but-k layout are known to every rank from the shard
    plan (no size exchange): ``x`` is this rank's piece flattened to (n_r, ...) rows with n_r = sizes[rank]; pieces are
    padded to max(sizes) rows.  Returns the list of the ranks' pieces (views, unpadded)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        assert len(sizes) == 1 and x.shape[0] == sizes[0]
        return [x]
    world = dist.get_world_size(group)
    assert len(sizes) == world and x.shape[0] == sizes[dist.get_rank(group)]
    mx = max(sizes)
    dev = x.device
    if dev.type == 'cuda' and dist.get_backend(group) == 'gloo':       # flow checks on a one-GPU box: gloo moves host memory
        x = x.cpu()
    buf = x.new_zeros((mx,) + tuple(x.shape[1:]))
    buf[:x.shape[0]].copy_(x)
    out = x.new_empty((world * mx,) + tuple(x.shape[1:]))
    dist.all_gather_into_tensor(out, buf, group=group)
    out = out.view((world, mx) + tuple(x.shape[1:])).to(dev)
    return [out[r, :sizes[r]] for r in range(world)]


def sharded_forward_2d(backend, vid_w, shallow_w, mask_full, grid, groups, rank, T, n_levels, texts, text_cls, tmasks, timings=None):
    """One video on a (query group) x (clip chunk) grid of ranks (``shard_plan_2d`` / ``make_grid_groups``).

    vid_w, shallow_w : this rank's window of the features (the whole video when the plan does not cut T)
    texts, text_cls, tmasks : ALL queries of the video (replicated, KB-sized); the rank takes its group's slice
    Returns the outputs of ALL queries on every rank, like ``model(..., eval=True)``.
    Collectives: AG-1 / AG-2 of ``sharded_forward`` inside the rank's clip-chunk group (skipped when T is not cut), then ONE
    all-gather of the packed full-length outputs across the query groups (static sizes: sum_l T_l rows x queries per group)."""
    ts, qs = grid['t_shards'], grid['q_groups']
    qg, t = rank // ts, rank % ts
    q_lo, q_hi = grid['queries'][qg]
    tg, qgs = groups if groups is not None else ([None] * qs, [None] * ts)
    mark = timings.mark if timings is not None else (lambda name: None)
    sub_t, sub_m, sub_c = list(texts[q_lo:q_hi]), list(tmasks[q_lo:q_hi]), text_cls[q_lo:q_hi]
    if ts > 1 and grid.get('hybrid') is not None:
        lg, of, mk = hybrid_forward(backend, vid_w, shallow_w, mask_full, grid['hybrid'], t, T, n_levels, sub_t, sub_c, sub_m,
                                    group=tg[qg], timings=timings)
    elif ts > 1:
        lg, of, mk = sharded_forward(backend, vid_w, shallow_w, mask_full, grid['plan'], t, T, n_levels, sub_t, sub_c, sub_m,
                                     group=tg[qg], timings=timings)
    else:
        mark('scores')
        correl = backend.scores(shallow_w, sub_c)
        mark('gate')
        gate = backend.gate(correl, mask_full)
        mark('forward')
        logits, offsets, masks = backend.forward_window(vid_w, shallow_w, mask_full, sub_t, sub_m, gate, T, 0)
        sizes = [T >> l for l in range(n_levels)]
        lg = [tuple(x[None] for x in logits[q].split(sizes)) for q in range(q_hi - q_lo)]
        of = [tuple(x[None] for x in offsets[q].split(sizes)) for q in range(q_hi - q_lo)]
        mk = [tuple(x[None] for x in masks[q].split(sizes)) for q in range(q_hi - q_lo)]
    if qs == 1:
        mark('done')
        return lg, of, mk
    # AG-3: the full-length outputs of this group's queries, packed (S, nq_g, 4), across the query groups
    mark('ag3')
    cat = lambda parts: torch.cat([torch.stack([parts[q][l][0] for q in range(len(parts))], 1) for l in range(n_levels)], 0)  # noqa: E731
    packed = torch.cat((cat(lg).unsqueeze(-1), cat(of), cat(mk).unsqueeze(-1).to(lg[0][0].dtype)), -1)      # (S, nq_g, 4)
    per = [b - a for a, b in grid['queries']]
    S = packed.shape[0]
    pieces = _gather_static(packed.transpose(0, 1).contiguous(), per, qgs[t])                 # (nq_g, S, 4) per group
    mark('unpack')
    full = torch.cat(pieces, 0)                                                               # (nq, S, 4)
    sizes = [T >> l for l in range(n_levels)]
    assert full.shape[1] == S == sum(sizes)
    out_l = [tuple(x[None] for x in full[q, :, 0].split(sizes)) for q in range(full.shape[0])]
    out_o = [tuple(x[None] for x in full[q, :, 1:3].split(sizes)) for q in range(full.shape[0])]
    out_m = [tuple(x[None] for x in (full[q, :, 3] != 0).split(sizes)) for q in range(full.shape[0])]
    mark('done')
    return out_l, out_o, out_m


def sharded_forward(backend, vid_w, shallow_w, mask_full, plan, rank, T, n_levels, texts, text_cls, tmasks, group=None,
                    timings=None):
    """Eval forward of ONE video sharded over the ranks of ``group``.

    vid_w, shallow_w : (D, w_hi - w_lo) this rank's window of the (zero padded) features
    mask_full        : (T,) bool validity of every clip of the whole video (cheap, replicated)
    plan             : ``shard_plan`` of ALL ranks (static: every rank derives every piece size from it); ``rank`` = ours
    timings          : optional dict; receives callables' wall-clock free event pairs (see bench.py --shard-T)
    Returns the whole video's outputs on every rank, exactly like ``model(..., eval=True)``:
    three lists (NQ) of L-tuples  logits (1, T_l), offsets (1, T_l, 2), masks (1, T_l).

    Exactly two collectives, both a single all-gather with plan-derived static sizes:
      AG-1  raw sidekick scores of the owned clips, (own, NQ) fp32;
      AG-2  the owned slice of every level packed as rows (logit, offset0, offset1, mask) fp32, (S_own, NQ, 4).
    """
    lo, hi, w_lo, w_hi = plan[rank]
    nq = text_cls.shape[0]
    mark = timings.mark if timings is not None else (lambda name: None)
    # AG-1: raw sidekick scores of the owned clips -> whole video on every rank
    own = shallow_w[:, lo - w_lo:hi - w_lo]
    mark('scores')
    sc = backend.scores(own, text_cls)                                                     # (nq, own)
    mark('ag1')
    pieces = _gather_static(sc.t().contiguous(), [p[1] - p[0] for p in plan], group)       # rank-major pieces (own_r, nq)
    correl = torch.cat(pieces, 0).t().contiguous()                                         # (nq, T)
    mark('gate')
    assert correl.shape == (nq, T)
    gate_full = backend.gate(correl, mask_full)                                            # identical on every rank
    gate_w = gate_full[:, w_lo:w_hi].contiguous()
    mark('forward')
    logits, offsets, masks = backend.forward_window(vid_w, shallow_w, mask_full[w_lo:w_hi].contiguous(), texts, tmasks,
                                                    gate_w, T, w_lo)
    mark('pack')
    # AG-2: owned slice of every level, packed
    Tw = w_hi - w_lo

    def owned_rows(x):                                  # (nq, S_w, ...) -> (S_own, nq, ...): levels concatenated
        parts, off = [], 0
        for l in range(n_levels):
            a, b = (lo - w_lo) >> l, (hi - w_lo) >> l
            parts.append(x[:, off + a:off + b])
            off += Tw >> l
        return torch.cat(parts, 1).transpose(0, 1)

    packed = torch.cat((owned_rows(logits).unsqueeze(-1), owned_rows(offsets), owned_rows(masks).unsqueeze(-1).to(logits.dtype)), -1)
    s_own = [sum((p[1] - p[0]) >> l for l in range(n_levels)) for p in plan]
    mark('ag2')
    pieces = _gather_static(packed.contiguous(), s_own, group)                             # (S_own_r, nq, 4) per rank
    mark('unpack')
    out_l, out_o, out_m = [], [], []
    offs = [0] * len(plan)
    for l in range(n_levels):
        lv = []
        for r, p in enumerate(plan):
            n = (p[1] - p[0]) >> l
            lv.append(pieces[r][offs[r]:offs[r] + n])
            offs[r] += n
        lv = torch.cat(lv, 0).transpose(0, 1)                                              # (nq, T_l, 4)
        out_l.append(lv[..., 0])
        out_o.append(lv[..., 1:3])
        out_m.append(lv[..., 3] != 0)
    mark('done')
    lg = [tuple(x[q][None] for x in out_l) for q in range(nq)]
    of = [tuple(x[q][None] for x in out_o) for q in range(nq)]
    mk = [tuple(x[q][None] for x in out_m) for q in range(nq)]
    return lg, of, mk


# ---------------------------------------------------------------------------------------------
# T-sharding WITHOUT recomputing the top of the pyramid: the pyramid cut at level k
# ---------------------------------------------------------------------------------------------
def arch_of(model) -> dict:
    """the layer counts the halos are functions of, read off the model the plan is for (modeling.PtTransformerEarlyFusionIterative):
    what ``shard_plan_2d(hybrid_arch=...)``, ``hybrid_plan(**arch)`` and ``receptive_field`` take"""
    n_embd_convs, n_stem, _ = model.vid_net.arch
    return dict(fusion_layers=len(model.fusion.layers), n_embd_convs=int(n_embd_convs), n_stem=int(n_stem), head_layers=int(model.head_layers))


def hybrid_halos(n_levels: int, win: int, k: int, *, fusion_layers: int, n_embd_convs: int, n_stem: int, head_layers: int):
    """(hA, hB): one-sided halo of the NARROW window in clips (levels 0..k: early fusion, embedding, the encoder up to level k, cls_head on
    the level-k grid, the refinement TCN's 2^L - 1 clips, the pooling chain down to level k, cls_head2 / reg_head) and of the COARSE window in
    level-k rows (it must cover the narrow window -- the TCN stacks the logits of every level at the narrow window's clips -- plus the
    encoder reach of levels k+1..L-1 on the level-k grid, the heads' three rows of the top level and the pooling chain), each rounded up to
    the window alignment of its pyramid (blocks.py:216: every level a multiple of win // 2).  The layer counts have no defaults: halos
    computed for another architecture are too small and the sharded outputs silently wrong (``arch_of(model)``)."""
    L, hw = n_levels, win // 2
    hw1 = max(hw, 1)
    LC = L - k
    r_enc = fusion_layers + n_embd_convs + n_stem * (1 + hw) + (1 + hw) + sum(2 ** (l - 1) + hw * 2 ** l for l in range(1, k + 1))
    hA = r_enc + 2 * (head_layers + 1) * 2 ** k + (2 ** L - 1) + 2 ** k
    an = 2 ** k * hw1
    hA = -(-hA // an) * an
    encC = sum(2 ** (j - 1) + hw * 2 ** j for j in range(1, LC))
    hB = hA // 2 ** k + encC + (head_layers + 2) * 2 ** (LC - 1)
    ac = 2 ** (LC - 1) * hw1
    hB = -(-hB // ac) * ac
    return hA, hB


def hybrid_plan(T: int, world: int, n_levels: int, win: int, k: int = None, **arch):
    """Per rank: owned clips [lo, hi), narrow window [n_lo, n_hi) (clips), coarse window [c_lo, c_hi) (level-k rows); ``k`` = None picks
    the split level with the fewest rows per rank.  Everything is a function of (T, world, L, w, arch): nothing is negotiated at run
    time.  Returns dict(k, ranks=[dict(lo, hi, n_lo, n_hi, c_lo, c_hi)], rows_factor)."""
    L = n_levels
    a = alignment(L, win)
    assert T % a == 0, f'T={T} must be a multiple of {a}'
    units = T // a

    def build(kk):
        hA, hB = hybrid_halos(L, win, kk, **arch)
        Tk = T >> kk
        ranks, worst, fits = [], 0, True
        for r in range(world):
            lo, hi = (units * r // world) * a, (units * (r + 1) // world) * a
            n_lo, n_hi = max(0, lo - hA), min(T, hi + hA)
            c_lo, c_hi = max(0, (lo >> kk) - hB), min(Tk, (hi >> kk) + hB)
            assert (n_lo >> kk) >= c_lo and (n_hi >> kk) <= c_hi
            rows = sum((n_hi - n_lo) >> l for l in range(kk + 1)) + sum((c_hi - c_lo) >> j for j in range(1, L - kk))
            worst = max(worst, rows)
            fits = fits and (c_hi - c_lo) <= (n_hi - n_lo)           # the coarse levels run in the narrow pyramid's scratch (dcf_hybrid_phase1)
            ranks.append(dict(lo=lo, hi=hi, n_lo=n_lo, n_hi=n_hi, c_lo=c_lo, c_hi=c_hi))
        even = sum((T // world) >> l for l in range(L))
        return dict(k=kk, ranks=ranks, rows_factor=worst / even, fits=fits, arch=dict(arch))

    if k is not None:
        p = build(k)
        assert p['fits'], f'split level {k}: a coarse window is longer than its narrow window (small shards): pick a higher level'
        return p
    cands = [p for p in (build(kk) for kk in range(1, L - 1)) if p['fits']]
    assert cands, f'no split level fits T={T} on {world} ranks'
    return min(cands, key=lambda p: (p['rows_factor'], p['k']))


def hybrid_forward(backend, vid_w, shallow_w, mask_full, plan, rank, T, n_levels, texts, text_cls, tmasks, group=None, timings=None):
    """Eval forward of ONE video sharded over the ranks of ``group`` with the pyramid cut at level ``plan['k']`` (``hybrid_plan``).

    vid_w, shallow_w : (D, n_hi - n_lo) this rank's NARROW window of the features
    Returns the whole video's outputs on every rank, exactly like ``model(..., eval=True)`` (and like ``sharded_forward``).

    Four collectives, each ONE all-gather with plan-derived static sizes:
      AG-1  raw sidekick scores of the owned clips (as ``sharded_forward``)
      AG-F  the owned slice of the level-k features, (own >> k, NQ, E) fp32       (8 MB in all at T = 65 536, k = 3, E = 256)
      AG-R  the owned slice of the refined level-k map, (own >> k, NQ, 32) fp32
      AG-2  the owned slice of every level's outputs, packed (logit, offset0, offset1, mask)
    """
    k, me = plan['k'], plan['ranks'][rank]
    ranks = plan['ranks']
    if hasattr(backend, 'arch'):                       # the halos of the plan are functions of the layer counts: they must be this model's
        assert backend.arch() == plan['arch'], f"the plan was made for {plan['arch']}, the model has {backend.arch()}"
    lo, hi, n_lo, n_hi, c_lo, c_hi = (me[x] for x in ('lo', 'hi', 'n_lo', 'n_hi', 'c_lo', 'c_hi'))
    nq = text_cls.shape[0]
    L = n_levels
    mark = timings.mark if timings is not None else (lambda name: None)
    # AG-1
    mark('scores')
    sc = backend.scores(shallow_w[:, lo - n_lo:hi - n_lo], text_cls)
    mark('ag1')
    pieces = _gather_static(sc.t().contiguous(), [p['hi'] - p['lo'] for p in ranks], group)
    correl = torch.cat(pieces, 0).t().contiguous()
    mark('gate')
    gate_full = backend.gate(correl, mask_full)
    # phase 1 + AG-F
    mark('phase1')
    featk_w = backend.hybrid_phase1(vid_w, shallow_w, mask_full[n_lo:n_hi].contiguous(), texts, tmasks,
                                    gate_full[:, n_lo:n_hi].contiguous(), T, n_lo, k, c_hi - c_lo)             # (nq, Tn >> k, E)
    own_k = [(p['hi'] - p['lo']) >> k for p in ranks]
    mark('agF')
    pieces = _gather_static(featk_w[:, (lo - n_lo) >> k:(hi - n_lo) >> k].transpose(0, 1).contiguous(), own_k, group)
    featk_c = torch.cat(pieces, 0)[c_lo:c_hi].transpose(0, 1).contiguous()                                        # (nq, Tc, E)
    # phase 2 + AG-R
    mark('phase2')
    maskk_c = mask_full[::2 ** k][c_lo:c_hi].contiguous()
    refk_w = backend.hybrid_phase2(featk_c, maskk_c, (n_lo >> k) - c_lo)                                          # (nq, Tn >> k, 32)
    mark('agR')
    pieces = _gather_static(refk_w[:, (lo - n_lo) >> k:(hi - n_lo) >> k].transpose(0, 1).contiguous(), own_k, group)
    refk_c = torch.cat(pieces, 0)[c_lo:c_hi].transpose(0, 1).contiguous()
    # phase 3
    mark('phase3')
    (lg_n, of_n, mk_n), (lg_c, of_c, mk_c) = backend.hybrid_phase3(refk_c)
    mark('pack')
    Tn, Tc = n_hi - n_lo, c_hi - c_lo

    def owned_rows(xn, xc):                     # -> (S_own, nq, ...): the owned slice of every level, levels concatenated
        parts, off = [], 0
        for l in range(k + 1):
            parts.append(xn[:, off + ((lo - n_lo) >> l):off + ((hi - n_lo) >> l)])
            off += Tn >> l
        off = 0
        for j in range(1, L - k):
            a_, b_ = ((lo >> k) - c_lo) >> j, ((hi >> k) - c_lo) >> j
            parts.append(xc[:, off + a_:off + b_])
            off += Tc >> j
        return torch.cat(parts, 1).transpose(0, 1)

    packed = torch.cat((owned_rows(lg_n, lg_c).unsqueeze(-1), owned_rows(of_n, of_c),
                        owned_rows(mk_n, mk_c).unsqueeze(-1).to(lg_n.dtype)), -1)
    s_own = [sum((p['hi'] - p['lo']) >> l for l in range(L)) for p in ranks]
    mark('ag2')
    pieces = _gather_static(packed.contiguous(), s_own, group)
    mark('unpack')
    out_l, out_o, out_m = [], [], []
    offs = [0] * len(ranks)
    for l in range(L):
        lv = []
        for r, p in enumerate(ranks):
            n = (p['hi'] - p['lo']) >> l
            lv.append(pieces[r][offs[r]:offs[r] + n])
            offs[r] += n
        lv = torch.cat(lv, 0).transpose(0, 1)
        out_l.append(lv[..., 0])
        out_o.append(lv[..., 1:3])
        out_m.append(lv[..., 3] != 0)
    mark('done')
    lg = [tuple(x[q][None] for x in out_l) for q in range(nq)]
    of = [tuple(x[q][None] for x in out_o) for q in range(nq)]
    mk = [tuple(x[q][None] for x in out_m) for q in range(nq)]
    return lg, of, mk
