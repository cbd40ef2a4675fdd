"""Where the time of forward + decode + NMS of one video goes (bench workload, default NMS config).  GPU only, dev tool."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('cvpr2025-decafnet_amd')
T = 16384
kw = dict(D=1024, E=256, TE=256, text_in=512, n_levels=8, win=9, n_heads=4, sn=60, sratio=0.3, msf=True, norm=True,
          max_seq_len=2304, text_layers=5, text_max_len=48, fusion_layers=2, max_vid_len=T)
opt = pkg.config.make_opt(**kw)
model = pkg.modeling.create_model(opt)
model.load_state_dict(pkg.synth.make_state_dict({k: list(v.shape) for k, v in model.state_dict().items()}, 2025))
model = model.cuda().eval()
model.reuse_output_buffers = True
inp = pkg.synth.make_inputs(1024, T, T, 1, 512, 32, 2028)
texts, tmasks = zip(*[model.encode_text(t[None].cuda(), torch.ones(1, 1, 32, dtype=torch.bool, device='cuda')) for t in inp['tokens']])
args = (inp['vid'].cuda(), inp['shallow_vid'].cuda(), inp['vid_masks'].cuda(), texts, inp['text_cls'].cuda(), tmasks)
ev = pkg.evaluator.GroundingEvaluator(opt, model)
nms = pkg.nms


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


model(*args, eval=True)
fl = model._last_flat
print('forward                     %.3f ms' % timed(lambda: model(*args, eval=True)))
print('collect_segments            %.3f ms' % timed(lambda: nms.collect_segments(*fl, T, 8)))
segs, scores, counts = nms.collect_segments(*fl, T, 8)
cfg = dict(opt.nms)
print('nms cfg', cfg, 'candidates', int(counts[0]))
print('batched_nms_queries         %.3f ms' % timed(lambda: nms.batched_nms_queries(segs, scores, counts, **cfg)))
print('  softnms_device max_iters  %.3f ms' % timed(lambda: nms.softnms_device(segs, scores, counts, 2000, 2000, cfg['iou_thresh'], cfg['sigma'], cfg['min_score'], 2, max_iters=cfg['max_num_segs'])))
d, _, oc = nms.softnms_device(segs, scores, counts, 2000, 2000, cfg['iou_thresh'], cfg['sigma'], cfg['min_score'], 2, max_iters=cfg['max_num_segs'])
kc = torch.clamp(oc, max=cfg['max_num_segs'])
top = d[:, :cfg['max_num_segs']].contiguous()
print('  voting_device             %.3f ms' % timed(lambda: nms.voting_device(top, kc, cfg['max_num_segs'], segs, scores, counts, 2000, cfg['voting_thresh'])))
print('generate_proposals          %.3f ms' % timed(lambda: ev.generate_proposals(fl, T, dict(fps=30.0, clip_stride=16, clip_size=32, duration=1e9))))
print('forward + proposals         %.3f ms' % timed(lambda: (model(*args, eval=True), ev.generate_proposals(model._last_flat, T, dict(fps=30.0, clip_stride=16, clip_size=32, duration=1e9)))))
