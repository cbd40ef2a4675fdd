"""Multi-process (gloo, world_size 2, CPU) coverage of the N > 1 paths: unit assignment, the shard
plan, and ``sharded_forward`` -- T-sharding with overlap-recompute halos and the two all-gathers --
driven by an oracle-backed compute backend.  The sharded outputs must equal the unsharded oracle."""
import importlib
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_pkg

sys.path.insert(0, ROOT)

KW = dict(D=32, E=32, TE=32, text_in=16, n_levels=4, win=5, n_heads=2, sn=8, sratio=0.3, msf=True, norm=True,
          max_seq_len=64, text_layers=1, text_max_len=24)
ARCH = dict(fusion_layers=2, n_embd_convs=2, n_stem=0, head_layers=2)     # the layer counts of the models below (dist.arch_of): the plans have no defaults
T, VID_LEN, NQ = 512, 470, 2


class OracleBackend:
    def __init__(self, sd, cfg):
        from oracle import decafnet_ref as R
        self.R, self.sd, self.cfg = R, sd, cfg

    def scores(self, shallow_own, text_cls):
        return self.R.sidekick_scores(shallow_own[None], text_cls, self.cfg['norm'])

    def gate(self, correl_full, mask_full):
        vl = int(mask_full.sum())
        out = torch.zeros_like(correl_full)
        for q in range(correl_full.shape[0]):
            out[q, :vl] = self.R.topk_block_gate(correl_full[q], vl, self.cfg['sn'], self.cfg['sratio'])
        return out

    def forward_window(self, vid_w, shallow_w, mask_w, texts, tmasks, gate_w, T_global, w_lo):
        R, cfg = self.R, self.cfg
        pe = R.position_encoding(cfg['vid_net']['max_seq_len'], cfg['vid_net']['embd_dim'])
        pe = R.resample_pe(pe, T_global, cfg['vid_net']['max_seq_len'])[:, w_lo:w_lo + vid_w.shape[-1]]
        lg, of, mk = R.forward_eval_window(self.sd, cfg, vid_w[None], shallow_w[None], mask_w[None], texts, tmasks, gate_w, pe)
        cat = lambda xs, d: torch.cat([torch.cat([x[0] for x in q], dim=0)[None] for q in xs], dim=0)  # noqa: E731
        return cat(lg, 0), cat(of, 0), cat(mk, 0)


class OracleHybridBackend(OracleBackend):
    """the three phases of dist.hybrid_forward on the CPU oracle: the specification the engine's dcf_hybrid_phase1 / 2 / 3 follow"""

    def hybrid_phase1(self, vid_w, shallow_w, mask_w, texts, tmasks, gate_w, T_global, n_lo, k, Tc):
        R, cfg, sd = self.R, self.cfg, self.sd
        vn = dict(cfg['vid_net'])
        pe = R.position_encoding(vn['max_seq_len'], vn['embd_dim'])
        pe = R.resample_pe(pe, T_global, vn['max_seq_len'])[:, n_lo:n_lo + vid_w.shape[-1]]
        vn['arch'] = (vn['arch'][0], vn['arch'][1], k + 1)
        self.k, self.L, self.narrow, self.texts = k, cfg['vid_net']['arch'][2], [], (texts, tmasks)
        feats = []
        for b in range(len(texts)):
            x = torch.cat([vid_w[None] * gate_w[b][None, None, :], shallow_w[None]], dim=1)
            m = mask_w[None, None, :]
            x, m = R.masked_conv1d(x, m, sd['vid_map.conv.weight'], sd['vid_map.conv.bias'])
            fused, fm = R.xattn_fusion(sd, cfg['fusion'], x, m, texts[b], tmasks[b])
            fpn, fpn_masks = R.video_transformer(sd, vn, fused, fm, pe_override=pe)
            self.narrow.append((list(fpn), list(fpn_masks)))
            feats.append(fpn[k][0].t())
        return torch.stack(feats, 0)                                     # (nq, Tn >> k, E)

    def hybrid_phase2(self, featk_c, maskk_c, off_k):
        R, cfg, sd, k, L = self.R, self.cfg, self.sd, self.k, self.L
        hl = cfg['cls_head'].get('n_layers', 2)
        vn = cfg['vid_net']
        self.coarse, refs = [], []
        for b in range(featk_c.shape[0]):
            x, m = featk_c[b].t()[None], maskk_c[None, None, :]
            fpn_c, masks_c = [x], [m]                                    # coarse level 0 = the gathered level-k features
            for j in range(1, L - k):
                x, m = R.transformer_encoder(sd, f'vid_net.branch.{k + j}', x, m, 2, vn['n_heads'], vn['mha_win_size'])
                fpn_c.append(x)
                masks_c.append(m)
            fpn_n, masks_n = self.narrow[b]
            l1n, _ = R.cls_head(sd, 'cls_head', fpn_n, masks_n, hl)
            l1c, _ = R.cls_head(sd, 'cls_head', fpn_c[1:], masks_c[1:], hl)
            Tn = l1n[0].shape[1]
            t = torch.arange(Tn)
            rows = [l1n[0][0]]
            for l in range(1, k + 1):
                rows.append(l1n[l][0][t >> l] * masks_n[0][0, 0])
            for j in range(1, L - k):
                idx = (((t >> k) + off_k) >> j).clamp(0, l1c[j - 1].shape[1] - 1)
                rows.append(l1c[j - 1][0][idx] * masks_n[0][0, 0])
            expand = R.tcn_refine(sd, 'refine', torch.stack(rows, 0)[None], masks_n[0], L)
            ref = [expand]
            for l in range(1, k + 1):
                ref.append(R.masked_max_pool1d(ref[-1], masks_n[l - 1])[0])
            self.narrow[b] = (fpn_n, masks_n, ref)
            self.coarse.append((fpn_c, masks_c))
            refs.append(ref[k][0].t())
        return torch.stack(refs, 0)                                      # (nq, Tn >> k, 32)

    def hybrid_phase3(self, refk_c):
        R, cfg, sd, k, L = self.R, self.cfg, self.sd, self.k, self.L
        hl = cfg['cls_head'].get('n_layers', 2)
        outs_n, outs_c = [], []
        for b in range(refk_c.shape[0]):
            fpn_n, masks_n, ref_n = self.narrow[b]
            fpn_c, masks_c = self.coarse[b]
            ref_c = [refk_c[b].t()[None]]
            for j in range(1, L - k):
                ref_c.append(R.masked_max_pool1d(ref_c[-1], masks_c[j - 1])[0])
            new_n = [torch.cat([f, r], 1) for f, r in zip(fpn_n, ref_n)]
            new_c = [torch.cat([f, r], 1) for f, r in zip(fpn_c[1:], ref_c[1:])]

            def heads(fpn, masks, first_level):
                lg, _ = R.cls_head(sd, 'cls_head2', fpn, masks, hl)
                raw = R.conv_head(sd, 'reg_head', 'reg_head', fpn, masks, cfg['reg_head'].get('n_layers', 2))
                of = [torch.relu(o * sd[f'reg_head.scales.{first_level + i}.scale']).transpose(1, 2) for i, o in enumerate(raw)]
                cat = lambda xs: torch.cat([x[0] for x in xs], 0)      # noqa: E731
                return cat(lg), cat(of), torch.cat([m[0, 0] for m in masks], 0)
            outs_n.append(heads(new_n, masks_n, 0))
            outs_c.append(heads(new_c, masks_c[1:], k + 1))
        st = lambda outs, i: torch.stack([o[i] for o in outs], 0)      # noqa: E731
        return (st(outs_n, 0), st(outs_n, 1), st(outs_n, 2)), (st(outs_c, 0), st(outs_c, 1), st(outs_c, 2))


def _worker_hybrid(rank, world, port, outdir, k):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        pkg, opt, sd, inp, texts, tmasks = _setup()
        d = pkg.dist
        plan = d.hybrid_plan(T, world, KW['n_levels'], KW['win'], k, **ARCH)
        me = plan['ranks'][rank]
        backend = OracleHybridBackend(sd, opt.model)
        with torch.no_grad():
            out = d.hybrid_forward(backend, inp['vid'][0][:, me['n_lo']:me['n_hi']], inp['shallow_vid'][0][:, me['n_lo']:me['n_hi']],
                                   inp['vid_masks'][0], plan, rank, T, KW['n_levels'], texts, inp['text_cls'], tmasks)
        torch.save((rank, [list(lv) for lv in out[0]], [list(lv) for lv in out[1]], [list(lv) for lv in out[2]], plan),
                   os.path.join(outdir, f'rank{rank}.pt'))
    finally:
        dist.destroy_process_group()


def _setup():
    pkg = load_pkg()
    opt = pkg.config.make_opt(**KW)
    model = pkg.modeling.create_model(opt)
    sd = pkg.synth.make_state_dict({k: list(v.shape) for k, v in model.state_dict().items()}, 5)
    inp = pkg.synth.make_inputs(KW['D'], T, VID_LEN, NQ, KW['text_in'], 6, 6)
    from oracle import decafnet_ref as R
    texts, tmasks = zip(*[R.encode_text(sd, opt.model, t[None], torch.ones(1, 1, t.size(-1), dtype=torch.bool)) for t in inp['tokens']])
    return pkg, opt, sd, inp, list(texts), list(tmasks)


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        pkg, opt, sd, inp, texts, tmasks = _setup()
        d = pkg.dist
        halo = d.receptive_field(KW['n_levels'], KW['win'], **d.PROBE_ARCH)
        plan = d.shard_plan(T, world, KW['n_levels'], KW['win'], halo)
        lo, hi, w_lo, w_hi = plan[rank]
        backend = OracleBackend(sd, opt.model)
        with torch.no_grad():
            out = d.sharded_forward(backend, inp['vid'][0][:, w_lo:w_hi], inp['shallow_vid'][0][:, w_lo:w_hi], inp['vid_masks'][0],
                                    plan, rank, T, KW['n_levels'], texts, inp['text_cls'], tmasks)
        torch.save((rank, [list(lv) for lv in out[0]], [list(lv) for lv in out[1]], [list(lv) for lv in out[2]], plan),
                   os.path.join(outdir, f'rank{rank}.pt'))
    finally:
        dist.destroy_process_group()


def _worker_2d(rank, world, port, outdir, nq, hybrid=False):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        pkg, opt, sd, inp, texts, tmasks = _setup()
        d = pkg.dist
        halo = d.receptive_field(KW['n_levels'], KW['win'], **d.PROBE_ARCH)
        grid = d.shard_plan_2d(T, world, nq, KW['n_levels'], KW['win'], halo, hybrid_arch=ARCH if hybrid else None)
        assert (grid['hybrid'] is not None) == (hybrid and grid['t_shards'] > 1)
        groups = d.make_grid_groups(grid['t_shards'], grid['q_groups'])
        t = rank % grid['t_shards']
        lo, hi, w_lo, w_hi = grid['plan'][t]
        backend = OracleHybridBackend(sd, opt.model) if hybrid else OracleBackend(sd, opt.model)
        with torch.no_grad():
            out = d.sharded_forward_2d(backend, inp['vid'][0][:, w_lo:w_hi], inp['shallow_vid'][0][:, w_lo:w_hi], inp['vid_masks'][0],
                                       grid, groups, rank, T, KW['n_levels'], texts[:nq], inp['text_cls'][:nq], tmasks[:nq])
        torch.save((rank, [list(lv) for lv in out[0]], [list(lv) for lv in out[1]], [list(lv) for lv in out[2]],
                    (grid['t_shards'], grid['q_groups'])), os.path.join(outdir, f'rank{rank}.pt'))
    finally:
        dist.destroy_process_group()


def test_shard_plan_2d_prefers_queries():
    """queries first, clips second: the rows a rank computes relative to an even share of T * NQ (BASELINE config 4 sizes)"""
    d = load_pkg().dist
    rf = d.receptive_field(8, 9, **d.PROBE_ARCH)
    want = {1: (8, 1, 1.5625), 2: (4, 2, 1.28125), 4: (2, 4, 1.0703125), 8: (1, 8, 1.0), 16: (1, 8, 1.0)}
    for nq, (ts, qs, factor) in want.items():
        g = d.shard_plan_2d(65536, 8, nq, 8, 9, rf)
        assert (g['t_shards'], g['q_groups']) == (ts, qs), (nq, g['t_shards'], g['q_groups'])
        assert abs(g['rows_factor'] - factor) < 0.02, (nq, g['rows_factor'])
        assert g['queries'][0][0] == 0 and g['queries'][-1][1] == nq
        assert all(a[1] == b[0] for a, b in zip(g['queries'], g['queries'][1:]))
    assert d.shard_plan_2d(65536, 8, 4, 8, 9, rf)['rows_factor'] <= 1.15      # VERDICT r02 item 5's bar, met from NQ = 4 on
    # with the pyramid cut allowed inside a clip-chunk group the NQ = 1 corner meets the bar too, and the plan's
    # windows are the narrow ones
    g1 = d.shard_plan_2d(65536, 8, 1, 8, 9, rf, hybrid_arch=ARCH)
    assert g1['t_shards'] == 8 and g1['hybrid'] is not None and g1['hybrid']['k'] == 3 and g1['rows_factor'] <= 1.15
    assert all(p[3] - p[2] <= 8192 + 2 * 384 for p in g1['plan']) and g1['plan'][3][3] - g1['plan'][3][2] == 8960
    g2 = d.shard_plan_2d(65536, 8, 2, 8, 9, rf, hybrid_arch=ARCH)
    assert (g2['t_shards'], g2['q_groups']) == (4, 2) and g2['hybrid'] is not None and g2['rows_factor'] <= 1.07
    assert d.shard_plan_2d(65536, 8, 8, 8, 9, rf, hybrid_arch=ARCH)['hybrid'] is None      # whole videos per rank: nothing to cut


@pytest.mark.timeout(900)
@pytest.mark.parametrize('world,nq,grid,hybrid', [(2, 2, (1, 2), False), (4, 2, (2, 2), False), (4, 2, (2, 2), True)])
def test_sharded_forward_2d_matches_unsharded(world, nq, grid, hybrid):
    """query groups x clip chunks over `world` gloo ranks: (2 ranks, 2 queries) is pure query sharding, (4 ranks, 2 queries) cuts T
    in two for each of two query groups; every rank must end with all queries' full-length outputs = the unsharded oracle"""
    import tempfile
    ctx = mp.get_context('spawn')
    port = 29700 + (os.getpid() + 7 * world + 3 * int(hybrid)) % 250
    with tempfile.TemporaryDirectory() as outdir:
        procs = [ctx.Process(target=_worker_2d, args=(r, world, port, outdir, nq, hybrid)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(800)
            assert p.exitcode == 0
        results = [torch.load(os.path.join(outdir, f'rank{r}.pt')) for r in range(world)]
    pkg, opt, sd, inp, texts, tmasks = _setup()
    from oracle import decafnet_ref as R
    with torch.no_grad():
        want = R.forward_eval(sd, opt.model, inp['vid'], inp['shallow_vid'], inp['vid_masks'], texts[:nq], inp['text_cls'][:nq], tmasks[:nq])
    for rank, lg, of, mk, g in results:
        assert g == grid
        assert len(lg) == nq
        for qi in range(nq):
            for l in range(KW['n_levels']):
                assert torch.equal(mk[qi][l], want[2][qi][l]), (rank, qi, l)
                torch.testing.assert_close(lg[qi][l], want[0][qi][l], rtol=1e-5, atol=2e-5)
                torch.testing.assert_close(of[qi][l], want[1][qi][l], rtol=1e-5, atol=2e-5)


def test_assign_units_balances():
    d = load_pkg().dist
    a = d.assign_units([32768, 2048, 4096, 16384, 8192, 2048, 30000, 1024], 4)
    assert sorted(sum(a, [])) == list(range(8))
    loads = [sum([32768, 2048, 4096, 16384, 8192, 2048, 30000, 1024][i] for i in r) for r in a]
    assert max(loads) == 32768 and min(loads) > 10000
    assert d.assign_units([5, 5], 4) == [[0], [1], [], []]


def test_shard_plan_alignment_and_cover():
    d = load_pkg().dist
    assert d.alignment(8, 9) == 512 and d.alignment(4, 5) == 16
    rf = d.receptive_field(8, 9, **d.PROBE_ARCH)
    assert rf == 2304                  # exact left reach; SURVEY 8e's probe saw the right reach, 2176 = rf - 2^(L-1)
    assert d.receptive_field(4, 5, **d.PROBE_ARCH) == 114 and d.receptive_field(5, 9, fusion_layers=1, n_embd_convs=0, n_stem=0, head_layers=1) == 253    # tools/receptive_field.py probes
    plan = d.shard_plan(65536, 8, 8, 9, rf)
    assert plan[0][0] == 0 and plan[-1][1] == 65536
    for (lo, hi, wl, wh), nxt in zip(plan, plan[1:] + [None]):
        assert lo % 512 == 0 and hi % 512 == 0 and wl % 128 == 0 and wh % 128 == 0 and (wh - wl) % 512 == 0
        assert wl <= lo - rf or wl == 0
        assert wh >= hi + rf or wh == 65536
        assert wh - wl <= (hi - lo) + 2 * 2304 + 511          # interior ranks: 12 800 clips for 8 192 owned (+56 %)
        if nxt:
            assert hi == nxt[0]


@pytest.mark.timeout(600)
def test_sharded_forward_matches_unsharded_world2():
    import tempfile
    ctx = mp.get_context('spawn')
    port = 29600 + os.getpid() % 300
    with tempfile.TemporaryDirectory() as outdir:
        procs = [ctx.Process(target=_worker, args=(r, 2, port, outdir)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(500)
            assert p.exitcode == 0
        results = [torch.load(os.path.join(outdir, f'rank{r}.pt')) for r in range(2)]
    pkg, opt, sd, inp, texts, tmasks = _setup()
    from oracle import decafnet_ref as R
    with torch.no_grad():
        want = R.forward_eval(sd, opt.model, inp['vid'], inp['shallow_vid'], inp['vid_masks'], texts, inp['text_cls'], tmasks)
    plan = results[0][4]
    assert plan[0][3] < T or plan[1][2] > 0, 'the test must really cut the video (windows smaller than T)'
    for rank, lg, of, mk, _ in results:
        for qi in range(NQ):
            for l in range(KW['n_levels']):
                assert torch.equal(mk[qi][l], want[2][qi][l]), (rank, qi, l)
                torch.testing.assert_close(lg[qi][l], want[0][qi][l], rtol=1e-5, atol=2e-5)
                torch.testing.assert_close(of[qi][l], want[1][qi][l], rtol=1e-5, atol=2e-5)


def test_hybrid_plan_rows():
    """the pyramid cut at level k: rows a rank computes over an even share (BASELINE config 4 sizes, one query), window alignments"""
    d = load_pkg().dist
    p = d.hybrid_plan(65536, 8, 8, 9, **ARCH)
    assert p['k'] == 3 and p['rows_factor'] <= 1.15, p['rows_factor']            # VERDICT r03 item 7's bar (the pure T-shard: 1.56)
    assert d.hybrid_plan(65536, 4, 8, 9, **ARCH)['rows_factor'] <= 1.07
    for world in (2, 4, 8):
        for k in (1, 2, 3, 4, 5):
            q = d.hybrid_plan(65536, world, 8, 9, k, **ARCH)
            prev = 0
            for r in q['ranks']:
                assert r['lo'] == prev and r['hi'] > r['lo']
                prev = r['hi']
                assert (r['n_hi'] - r['n_lo']) % (4 << k) == 0 and r['n_lo'] % (1 << k) == 0
                assert (r['c_hi'] - r['c_lo']) % (4 << (8 - k - 1)) == 0 and r['c_lo'] % (1 << (8 - k - 1)) == 0
                assert (r['n_lo'] >> k) >= r['c_lo'] and (r['n_hi'] >> k) <= r['c_hi']
            assert prev == 65536


@pytest.mark.timeout(900)
@pytest.mark.parametrize('world,k', [(2, 1), (2, 2), (4, 1)])
def test_hybrid_forward_matches_unsharded(world, k):
    """hybrid_forward over `world` gloo ranks (pyramid cut at level k: narrow windows for levels <= k, coarse windows above, four
    static-size all-gathers): every rank ends with the whole video's outputs = the unsharded oracle"""
    import tempfile
    ctx = mp.get_context('spawn')
    port = 29300 + (os.getpid() + 13 * world + k) % 250
    with tempfile.TemporaryDirectory() as outdir:
        procs = [ctx.Process(target=_worker_hybrid, args=(r, world, port, outdir, k)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(800)
            assert p.exitcode == 0
        results = [torch.load(os.path.join(outdir, f'rank{r}.pt')) for r in range(world)]
    pkg, opt, sd, inp, texts, tmasks = _setup()
    from oracle import decafnet_ref as R
    with torch.no_grad():
        want = R.forward_eval(sd, opt.model, inp['vid'], inp['shallow_vid'], inp['vid_masks'], texts, inp['text_cls'], tmasks)
    plan = results[0][4]
    assert any(r['n_hi'] - r['n_lo'] < T for r in plan['ranks']), 'the test must really cut the video'
    for rank, lg, of, mk, _ in results:
        for qi in range(NQ):
            for l in range(KW['n_levels']):
                assert torch.equal(mk[qi][l], want[2][qi][l]), (rank, qi, l)
                torch.testing.assert_close(lg[qi][l], want[0][qi][l], rtol=1e-5, atol=2e-5)
                torch.testing.assert_close(of[qi][l], want[1][qi][l], rtol=1e-5, atol=2e-5)
