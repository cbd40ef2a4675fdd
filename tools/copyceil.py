"""Copy-kernel ceiling beside the cross-attention core on the same rotating buffers (torch copy: ~5.3 TB/s on MI355X)."""
import torch, json, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import xattn_bench
def copy(rows, E, reps=100):
    per_set = 2*rows*E*4
    nsets = max(2, (600*2**20)//per_set + 1)
    a = [torch.randn(rows, E, device='cuda') for _ in range(nsets)]
    b = [torch.empty(rows, E, device='cuda') for _ in range(nsets)]
    for _ in range(5): b[0].copy_(a[0])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): b[i%nsets].copy_(a[i%nsets])
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1)*1e3/reps
    return us, per_set/us/1e3
for i in range(3):
    print('copy', copy(32768, 1024))
    r = xattn_bench.run(4096, 1024, 16, B=8)
    print('xattn', r['cold']['us'], r['cold']['frac_hbm_peak'], r['warm']['frac_hbm_peak'])
