"""What the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on the level-0 shapes of the five-video forward, in fp16 and
fp32: context for the f16x3 numbers (three fp16 MFMA products per fp32-accurate multiply-add, fp32 operands in memory).  GPU only."""
import torch, time
shapes = [(81920, 256, 256), (81920, 1024, 256), (81920, 256, 1024), (163200, 288, 864), (81920, 512, 256)]
def rate(M, N, K, dt):
    a = torch.randn(M, K, device='cuda', dtype=dt); w = torch.randn(N, K, device='cuda', dtype=dt)
    for _ in range(3): (a @ w.t())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): c = a @ w.t()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    return us, 2.0 * M * N * K / us / 1e6
torch.backends.cuda.matmul.allow_tf32 = False
for M, N, K in shapes:
    r = {str(dt).split('.')[-1]: rate(M, N, K, dt) for dt in (torch.float16, torch.bfloat16, torch.float32)}
    print(f'{M}x{N}x{K}: ' + '  '.join(f'{k} {v[0]:.1f} us {v[1]:.0f} TFLOP/s' for k, v in r.items()))
