import sys, os, time, importlib
sys.path.insert(0, os.getcwd())
import torch, bench
pkg = importlib.import_module('cvpr2025-decafnet_amd')
T = 16384
kw = bench.probe_kwargs(T)
opt = pkg.config.make_opt(**kw); opt.model['max_batch'] = 8
dev = torch.device('cuda', 0)
m0 = pkg.modeling.create_model(opt)
sd = pkg.synth.make_state_dict({k: list(v.shape) for k, v in m0.state_dict().items()}, 2025)
def make(seed, nq):
    m = pkg.modeling.create_model(opt); m.load_state_dict(sd); m = m.to(dev).eval().requires_grad_(False); m.reuse_output_buffers = True
    inp = pkg.synth.make_inputs(kw['D'], T, T, nq, kw['text_in'], 32, seed)
    tx, tm = zip(*[m.encode_text(t[None].to(dev), torch.ones(1, 1, 32, dtype=torch.bool, device=dev)) for t in inp['tokens']])
    return m, (inp['vid'].to(dev), inp['shallow_vid'].to(dev), inp['vid_masks'].to(dev), tx, inp['text_cls'].to(dev), tm)
for nq in (1, 8):
    for ns in (1, 2, 3, 4):
        ms = [make(100 + i, nq) for i in range(ns)]
        ss = [torch.cuda.Stream() for _ in range(ns)]
        def step():
            for (m, a), s in zip(ms, ss):
                with torch.cuda.stream(s):
                    m(*a, eval=True)
        for _ in range(4): step()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
        print(f'nq={nq} streams={ns}: {ns * nq * T / dt / 1e6:.2f} M clips/s  {1e3 * dt:.3f} ms per step')
        del ms
