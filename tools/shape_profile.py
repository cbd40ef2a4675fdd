"""Per-shape GEMM timing of one forward (DCF_PROF_SHAPES=1 labels): python tools/shape_profile.py [T] [nq]"""
import ctypes, importlib, json, os, sys
os.environ['DCF_PROF_SHAPES'] = '1'
os.environ['DCF_NO_GRAPH'] = '1'
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

T = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1
pkg = importlib.import_module('cvpr2025-decafnet_amd')
lib = pkg._lib.lib()
kw = bench.probe_kwargs(T)
opt = pkg.config.make_opt(**kw)
opt.model['max_batch'] = 8
model = pkg.modeling.create_model(opt)
sd = pkg.synth.make_state_dict({k: list(v.shape) for k, v in model.state_dict().items()}, 2025)
model.load_state_dict(sd)
model = model.cuda().eval().requires_grad_(False)
inp = pkg.synth.make_inputs(kw['D'], T, T, nq, kw['text_in'], 32, 2028)
vid, shallow, vmask = inp['vid'].cuda(), inp['shallow_vid'].cuda(), inp['vid_masks'].cuda()
texts, tmasks = [], []
for tok in inp['tokens']:
    t, m = model.encode_text(tok[None].cuda(), torch.ones(1, 1, tok.size(-1), dtype=torch.bool, device='cuda'))
    texts.append(t); tmasks.append(m)
step = lambda: model(vid, shallow, vmask, tuple(texts), inp['text_cls'].cuda(), tuple(tmasks), eval=True)
for _ in range(5): step()
torch.cuda.synchronize()
lib.dcf_profile_enable(1)
N = 10
for _ in range(N): step()
torch.cuda.synchronize()
need = lib.dcf_profile_report(None, 0)
buf = ctypes.create_string_buffer(int(need) + 16)
lib.dcf_profile_report(buf, len(buf))
prof = json.loads(buf.value.decode())
tot = sum(v['ms'] for v in prof.values()) / N
print('total event ms/step', tot)
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms']):
    ms = v['ms'] / N
    print(f"{k:60s} n={v['count']//N:3d} {1e3*ms:8.1f} us  {1e3*ms/(v['count']//N):7.1f} us/launch  {v['flops']/N/(ms*1e-3)/1e12 if ms>0 else 0:6.1f} TF")
