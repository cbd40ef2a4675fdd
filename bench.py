#!/usr/bin/env python3
"""Throughput of the DeCafNet grounding forward on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--nq NQ] [--T 16384]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one eval forward of PtTransformerEarlyFusionIterative over one synthetic video of
T=16384 clips (D=1024) for NQ text queries, inputs already resident in HBM, timed from
model.forward entry to logits/offsets/masks on the device.  Workload =
BASELINE.json configs[2] with the survey's probe hyper-parameters.
With N > 1 every rank runs its own (video, queries) replica -- the path shards over independent
(video, query) units with no data-path collective (weak scaling); value = all clips of all ranks
divided by the slowest rank's time.

Rank 0 prints ONE JSON line: the driver contract fields plus
  roofline     : dominant kernel (fp32 MFMA GEMM), live HIP-event timing of every launch
  cpu_baseline : the CPU oracle (a port of the reference algorithm) timed on this host's cores
  stages       : per-kernel-family time of one step (same event timing)
  post         : proposal decode + NMS latency and the NMS index match against the oracle
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: bf16 dense MFMA peak
PEAK_HBM_GBS = 8000.0            # HBM3E spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--nq', type=int, default=1, help='text queries per video')
    ap.add_argument('--T', type=int, default=16384)
    ap.add_argument('--vid-len', type=int, default=0, help='valid clips (default: T)')
    ap.add_argument('--max-batch', type=int, default=8)
    ap.add_argument('--videos', type=int, default=3,
                    help='videos per step: independent videos in flight on their own HIP streams (one model instance each)')
    ap.add_argument('--batch', type=int, default=5,
                    help='videos per forward (forward_videos: same-length videos batched through every kernel); a step then '
                         'holds --videos x --batch videos')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--debug-gloo-one-gpu', action='store_true',
                    help='flow check of the multi-rank path on a one-GPU box: gloo rendezvous, every rank on cuda:0 (timings meaningless)')
    ap.add_argument('--no-post', action='store_true')
    ap.add_argument('--cpu-T', type=int, default=0, help='T of the CPU-baseline sample (default: same T)')
    return ap.parse_args()


def probe_kwargs(T):
    # BASELINE.md section 2 probe hyper-parameters; max_seq_len*10 >= T so PtGenerator covers the video
    return dict(D=1024, E=256, TE=256, text_in=512, n_levels=8, win=9, n_heads=4, sn=60, sratio=0.3, msf=True,
                norm=True, max_seq_len=2304, text_layers=5, text_max_len=48, fusion_layers=2)


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        if args.debug_gloo_one_gpu:
            dist.init_process_group('gloo')
            local_rank = 0
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    assert torch.cuda.is_available(), 'bench.py needs an MI355X'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    pkg = importlib.import_module('cvpr2025-decafnet_amd')
    lib = pkg._lib.lib()

    T = args.T
    vid_len = args.vid_len or T
    kw = probe_kwargs(T)
    opt = pkg.config.make_opt(**kw)
    opt.model['max_batch'] = args.max_batch
    model = pkg.modeling.create_model(opt)
    shapes = {k: list(v.shape) for k, v in model.state_dict().items()}
    sd = pkg.synth.make_state_dict(shapes, 2025)
    model.load_state_dict(sd)
    model = model.to(dev).eval().requires_grad_(False)
    model.reuse_output_buffers = True      # static output buffers: the repeated forward replays one captured HIP graph
    inp = pkg.synth.make_inputs(kw['D'], T, vid_len, args.nq, kw['text_in'], 32, 2025 + 3 + rank)
    vid, shallow, vmask = inp['vid'].to(dev), inp['shallow_vid'].to(dev), inp['vid_masks'].to(dev)
    text_cls = inp['text_cls'].to(dev)
    texts, tmasks = [], []
    for tok in inp['tokens']:
        t, m = model.encode_text(tok[None].to(dev), torch.ones(1, 1, tok.size(-1), dtype=torch.bool, device=dev))
        texts.append(t)
        tmasks.append(m)
    texts, tmasks = tuple(texts), tuple(tmasks)

    def lane_inputs(mdl, seed0):
        """the --batch - 1 further videos of a lane (its first video is given): argument tuples of forward"""
        extra = []
        for j in range(1, max(1, args.batch)):
            ij = pkg.synth.make_inputs(kw['D'], T, vid_len, args.nq, kw['text_in'], 32, seed0 + 100000 * j)
            tj, mj = zip(*[mdl.encode_text(tok[None].to(dev), torch.ones(1, 1, tok.size(-1), dtype=torch.bool, device=dev))
                           for tok in ij['tokens']])
            extra.append((ij['vid'].to(dev), ij['shallow_vid'].to(dev), ij['vid_masks'].to(dev), tuple(tj), ij['text_cls'].to(dev), tuple(mj)))
        return extra

    first = (vid, shallow, vmask, texts, text_cls, tmasks)
    batch0 = [first] + lane_inputs(model, 2025 + 3 + rank)

    def step1():
        if args.batch > 1:
            return model.forward_videos(batch0)
        return model(vid, shallow, vmask, texts, text_cls, tmasks, eval=True)

    # A step = one batch of `--videos` different synthetic videos, each an independent forward (own model instance =
    # own workspace + HIP graph) on its own stream.  One forward is a chain of ~110 dependent kernels, about half of
    # them single-round GEMMs that leave CUs idle in their prologue / epilogue; independent videos fill those holes
    # (NQ = 1: 7.1 / 8.6 / 9.2 / 8.6-9.4 M clips/s with 1 / 2 / 3 / 4 videos in flight, tools/streams_probe.py; the fourth
    # stream shares a hardware queue on some runs, so three is the default).
    others = []
    for k in range(1, max(1, args.videos)):
        mk = model.replica()                        # same parameters, own engine / workspace / graph
        mk.reuse_output_buffers = True
        ik = pkg.synth.make_inputs(kw['D'], T, vid_len, args.nq, kw['text_in'], 32, 2025 + 3 + rank + 1000 * k)
        tk, mkk = zip(*[mk.encode_text(tok[None].to(dev), torch.ones(1, 1, tok.size(-1), dtype=torch.bool, device=dev))
                        for tok in ik['tokens']])
        ak = (ik['vid'].to(dev), ik['shallow_vid'].to(dev), ik['vid_masks'].to(dev), tuple(tk), ik['text_cls'].to(dev), tuple(mkk))
        others.append((mk, ak, torch.cuda.Stream(), [ak] + lane_inputs(mk, 2025 + 3 + rank + 1000 * k)))
    stream0 = torch.cuda.Stream() if others else None
    torch.cuda.synchronize()

    def step():
        if not others:
            return step1()
        with torch.cuda.stream(stream0):
            out0 = step1()
        for mk, a, sk, bk in others:
            with torch.cuda.stream(sk):
                if args.batch > 1:
                    mk.forward_videos(bk)
                else:
                    mk(*a, eval=True)
        return out0

    # setup, not steps: the engine runs a new argument set eagerly once, captures its HIP graph on the second call and
    # replays from the third; a few more replays let the allocator and the clocks settle whatever --warmup is
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device='cpu' if args.debug_gloo_one_gpu else dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # the f16x3 GEMMs flag any accumulator that left the finite range (operands beyond the fp16 range): must be clean
    for mk in [model] + [o[0] for o in others]:
        assert mk.numerics_status() & 1 == 0, 'f16x3 GEMM range overflow flagged: the timed outputs are not valid'
    n_lanes = 1 + len(others)
    n_videos = n_lanes * max(1, args.batch)
    clips_per_step = n_videos * vid_len * args.nq
    value = world * clips_per_step * args.steps / elapsed

    result = {
        'metric': 'clips/sec (grounding fwd, T=16384 D=1024)', 'value': value, 'unit': 'clips/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'BASELINE configs[2]: T={T} D=1024 full multi-scale pyramid + sidekick top-k 30% + expert path, '
                               f'NQ={args.nq} queries/video, {n_videos} independent videos per step: {n_lanes} forwards in flight on {n_lanes} HIP streams x '
                               f'{max(1, args.batch)} videos per forward, one replica of this per GPU',
                   'T': T, 'vid_len': vid_len, 'D': 1024, 'E': 256, 'TE': 256, 'levels': 8, 'win': 9, 'heads': 4,
                   'fusion_layers': 2, 'sn': 60, 'sratio': 0.3, 'msf': True, 'norm': True, 'Lq': 32, 'nq': args.nq,
                   'max_batch': args.max_batch, 'videos_per_step': n_videos, 'forwards_in_flight': n_lanes, 'videos_per_forward': max(1, args.batch), 'parallelism': f'replicas x{world}',
                   'launch': 'HIP graph replay of the forward (captured on the 2nd identical call); DCF_NO_GRAPH=1 = eager'},
    }

    if rank == 0:
        # ---- live per-kernel timing: the same steps again with every launch bracketed by HIP events
        lib.dcf_profile_enable(1)
        for _ in range(args.steps):
            step1()                                  # one video on the current stream: undisturbed per-kernel times
        torch.cuda.synchronize()
        need = lib.dcf_profile_report(None, 0)
        buf = ctypes.create_string_buffer(int(need) + 16)
        lib.dcf_profile_report(buf, len(buf))
        lib.dcf_profile_enable(0)
        prof = json.loads(buf.value.decode())
        tot_ms = sum(v['ms'] for v in prof.values())
        stages = {k: {'ms_per_step': v['ms'] / args.steps, 'launches_per_step': v['count'] / args.steps,
                      'share': v['ms'] / tot_ms,
                      'tflops': (v['flops'] / (v['ms'] * 1e-3) / 1e12) if v['ms'] > 0 else 0.0,
                      'alg_GBps': (v['bytes'] / (v['ms'] * 1e-3) / 1e9) if v['ms'] > 0 else 0.0}
                  for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms'])}
        # dominant kernel = the dense-conv GEMM.  Default arithmetic: fp32-accurate bf16x6 operand split on
        # v_mfma_f32_32x32x16_bf16 (6 bf16 MFMA products per fp32 MAC) => fp32-equivalent peak = 2500 / 6 TFLOP/s;
        # with DCF_GEMM_MODE=fp32 the native fp32 MFMA (157.3 TFLOP/s) is used instead.
        # f16x3 (default): two fp16 planes per operand, 3 fp16 MFMA products per fp32 MAC => peak 2500 / 3 TFLOP/s; the two
        # vid_map GEMMs on the raw features stay on bf16x6 and are listed under `stages`
        fam = 'gemm_f16x3' if any(k.startswith('gemm_f16x3') for k in prof) else \
              ('gemm_bf16x6' if any(k.startswith('gemm_bf16x6') for k in prof) else 'gemm_f32')
        gk = [k for k in prof if k.startswith(fam)]
        d = {f: sum(prof[k][f] for k in gk) for f in ('ms', 'flops', 'bytes', 'count')}
        terms = {'gemm_bf16x6': 6, 'gemm_f16x3': 3, 'gemm_f32': 0}[fam]
        peak = PEAK_BF16_MFMA_TFLOPS / terms if terms else PEAK_F32_MFMA_TFLOPS
        ach = d['flops'] / (d['ms'] * 1e-3) / 1e12
        result['roofline'] = {
            'kernel': '%s_kernel<*> (all %d tile/operand instantiations, %.0f%% of the step)' % (
                'gemm_bf16s' if terms else 'gemm_f32', len(gk), 100 * d['ms'] / tot_ms),
            'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak, 'traffic': None,
            'launches': d['count'], 'avg_launch_us': 1e3 * d['ms'] / d['count'],
            'alg_flops_per_launch': d['flops'] / d['count'],
            'peak_note': ('fp16 / bf16 dense MFMA peak 2500 TFLOP/s / %d 16-bit MFMA products per fp32 multiply-add (fp32-accurate operand split)' % terms)
                         if terms else 'native fp32 MFMA dense peak',
            'frac_of_native_f32_mfma_peak': ach / PEAK_F32_MFMA_TFLOPS,
            'note': 'HIP events around every launch of this kernel, same K steps re-run right after the timed region; '
                    'achieved = algorithmic 2*M*N*K flops / event time',
        }
        # HBM-side bytes per launch of this kernel family: rocprofv3 PMC passes cannot run inside this process, so the
        # figure comes from the committed summary of tools/pmc_traffic.sh (same command, same build); null if absent
        try:
            # the committed summary was collected on the DEFAULT workload: only a default run may quote it
            if (args.T, args.nq, args.videos, args.batch, args.vid_len) != (16384, 1, 3, 5, 0):
                raise KeyError('non-default workload')
            with open(os.path.join(ROOT, 'profiles', 'r01_pmc_gemm_traffic.json')) as fh:
                pmc = json.load(fh).get({6: 'gemm_bf16s', 3: 'gemm_f16x3', 0: 'gemm_f32'}[terms])
            if pmc:
                result['roofline']['traffic'] = pmc['hbm_bytes_per_launch']
                # the other roofline of the same kernels: with f16x3 the fp32 activations are the larger cost
                gbps = pmc['hbm_bytes_per_launch'] / (1e-6 * result['roofline']['avg_launch_us']) / 1e9
                result['roofline']['hbm_view'] = {'achieved': gbps, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': gbps / PEAK_HBM_GBS,
                                                  'note': 'HBM-side bytes per launch (PMC) / average launch time of the GEMM family'}
                result['roofline']['traffic_note'] = ('bytes per launch, FETCH_SIZE x2 + WRITE_SIZE from two rocprofv3 --pmc passes '
                                                      '(profiles/r01_pmc_gemm_traffic.json, tools/pmc_traffic.sh); algorithmic '
                                                      'operand bytes per launch: %.3e' % (d['bytes'] / d['count']))
        except (OSError, ValueError, KeyError):
            pass
        result['config']['gemm_mode'] = {6: 'bf16x6 split MFMA (fp32 accurate)', 3: 'f16x3 split MFMA (fp32 accurate)', 0: 'native fp32 MFMA'}[terms]
        xa = prof.get('xattn_core')
        if xa:
            result['xattn_in_forward'] = {'bound': 'hbm', 'achieved': xa['bytes'] / (xa['ms'] * 1e-3) / 1e9, 'peak': PEAK_HBM_GBS,
                                          'unit': 'GB/s', 'frac': xa['bytes'] / (xa['ms'] * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                          'avg_launch_us': 1e3 * xa['ms'] / xa['count'],
                                          'note': 'warm: the 16 MiB q/ctx tensors of one query live in L2/Infinity Cache'}
        # BASELINE config 2 (T=4096, E=1024, Lk=33 cross-attention core), measured as SURVEY 8d prescribes: 100
        # back-to-back launches over rotating buffers > 512 MB, 8 queries per launch (268 MB of q/ctx traffic)
        extras = not args.no_post and world == 1          # single-GPU run only: the other ranks would wait at the barrier
        if extras:
            sys.path.insert(0, os.path.join(ROOT, 'tools'))
            import xattn_bench
            # run twice, keep the second: the first call works on ~800 MB of freshly hipMalloc'ed buffers and is 20 % slower
            # for all of its 140 launches (73 vs 61 us, tools/xattn_bench.py run back to back), the second reuses the cached blocks
            xattn_bench.run(4096, 1024, 16, Lk=33, B=8, reps=100)
            x = xattn_bench.run(4096, 1024, 16, Lk=33, B=8, reps=100)
            result['xattn_config2'] = {'bound': 'hbm', 'achieved': x['cold']['GBps'], 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                                       'frac': x['cold']['GBps'] / PEAK_HBM_GBS, 'warm_GBps': x['warm']['GBps'],
                                       'us_per_launch': x['cold']['us'], 'alg_bytes_per_clip': 8 * 1024,
                                       'workload': 'T=4096 E=1024 heads=16 Lk=33, 8 queries/launch, fp32 MFMA 16x16x4'}
        result['stages'] = stages
        result['event_ms_per_step'] = tot_ms / args.steps

        # ---- the same video with 8 queries per forward (how Evaluator calls the model: all queries of a video at once)
        if extras and args.nq == 1:
            inp8 = pkg.synth.make_inputs(kw['D'], T, vid_len, 8, kw['text_in'], 32, 2025 + 11)
            cls8 = inp8['text_cls'].to(dev)
            t8, m8 = zip(*[model.encode_text(tok[None].to(dev), torch.ones(1, 1, tok.size(-1), dtype=torch.bool, device=dev))
                           for tok in inp8['tokens']])
            for _ in range(3):
                model(vid, shallow, vmask, t8, cls8, m8, eval=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                model(vid, shallow, vmask, t8, cls8, m8, eval=True)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / 5
            result['nq8'] = {'value': vid_len * 8 / dt, 'unit': 'clips/s', 'ms_per_step': 1e3 * dt,
                             'note': 'one forward over the same video with 8 queries (batched through every kernel), rank 0 only'}
            model(vid, shallow, vmask, texts, text_cls, tmasks, eval=True)      # restore _last_flat for the post-processing leg

        # ---- the same workload with ONE video in flight (latency view of the headline)
        if extras and others:
            for _ in range(3):
                step1()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step1()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / args.steps
            result['one_video_in_flight'] = {'value': max(1, args.batch) * vid_len * args.nq / dt, 'unit': 'clips/s', 'ms_per_forward': 1e3 * dt,
                                             'note': 'a single forward at a time on one stream (HIP-graph replay), rank 0 only'
                                                     + (f'; {args.batch} videos per forward' if args.batch > 1 else '')}

        # ---- proposal decode + NMS (reported separately) and the NMS index match
        if extras:
            from oracle import nms_oracle
            fl, fo, fm = model._last_flat
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                segs, scores, counts = pkg.nms.collect_segments(fl, fo, fm, T, 8)
            torch.cuda.synchronize()
            t_collect = (time.perf_counter() - t1) / reps
            n = int(counts[0])
            s_d, c_d = segs[:1, :n].contiguous(), scores[:1, :n].contiguous()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                keep, kc = pkg.nms.nms_device(s_d, c_d, None, n, n, 0.5)
            torch.cuda.synchronize()
            t_nms = (time.perf_counter() - t1) / reps
            t1 = time.perf_counter()
            for _ in range(reps):
                dets, inds, oc = pkg.nms.softnms_device(s_d, c_d, None, n, n, 0.1, 0.9, 0.001, 2)
            torch.cuda.synchronize()
            t_soft = (time.perf_counter() - t1) / reps
            s_cpu, c_cpu = s_d[0].cpu(), c_d[0].cpu()
            ref_keep = nms_oracle.nms(s_cpu, c_cpu, 0.5)
            d2 = torch.zeros(n, 3)
            ref_soft = nms_oracle.softnms(s_cpu, c_cpu, d2, 0.1, 0.9, 0.001, 2)
            t1 = time.perf_counter()
            nms_oracle.nms(s_cpu, c_cpu, 0.5)
            t_cpu_nms = time.perf_counter() - t1
            result['post'] = {
                'candidates': n, 'collect_ms_per_video': 1e3 * t_collect, 'nms_ms': 1e3 * t_nms, 'softnms_full_ms': 1e3 * t_soft,
                'cpu_oracle_nms_ms': 1e3 * t_cpu_nms,
                'nms_index_match': bool(torch.equal(keep[0, :int(kc)].cpu(), ref_keep)),
                'softnms_index_match': bool(torch.equal(inds[0, :int(oc)].cpu(), ref_soft)),
            }

        # ---- CPU baseline: the oracle (port of the reference algorithm) on this host, bounded sample
        if not args.no_cpu_baseline and world == 1:      # rank 0 at N = 1 only
            from oracle import decafnet_ref as R
            cpu_T = args.cpu_T or T
            cinp = pkg.synth.make_inputs(kw['D'], cpu_T, min(vid_len, cpu_T), 1, kw['text_in'], 32, 2025 + 3)
            t_cpu, m_cpu = R.encode_text(sd, opt.model, cinp['tokens'][0][None], torch.ones(1, 1, 32, dtype=torch.bool))
            with torch.no_grad():
                t1 = time.perf_counter()
                R.forward_eval(sd, opt.model, cinp['vid'], cinp['shallow_vid'], cinp['vid_masks'], [t_cpu], cinp['text_cls'], [m_cpu])
                cpu_s = time.perf_counter() - t1
            result['cpu_baseline'] = {
                'value': min(vid_len, cpu_T) / cpu_s, 'unit': 'clips/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                'sample': f'1 video x 1 query, T={cpu_T}, oracle/decafnet_ref.py forward_eval (torch {torch.__version__} CPU fp32), {cpu_s:.2f} s',
            }
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
