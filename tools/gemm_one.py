"""Run one GEMM shape repeatedly (for rocprofv3 --pmc).  usage: gemm_one.py MODE M N K [reps]   MODE = f32|x6|f16"""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
act = int(sys.argv[6]) if len(sys.argv) > 6 else 0
pkg = importlib.import_module('cvpr2025-decafnet_amd')
lib = pkg._lib.lib()
A = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda'); b = torch.randn(N, device='cuda')
C = torch.empty(M, N, device='cuda')
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
import time, json
for _ in range(3):
    lib.dcf_op_linear_split(P(A), P(W), P(b), P(C), M, N, K, act, 6 if mode == 'x6' else 16, st) if mode != 'f32' else lib.dcf_op_linear(P(A), P(W), P(b), P(C), M, N, K, act, st)
torch.cuda.synchronize()
lib.dcf_profile_enable(1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    if mode == 'f32':
        lib.dcf_op_linear(P(A), P(W), P(b), P(C), M, N, K, act, st)
    else:
        lib.dcf_op_linear_split(P(A), P(W), P(b), P(C), M, N, K, act, 6 if mode == 'x6' else 16, st)
torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
print(mode, M, N, K, 'us/launch (incl. weight split for x6/x3):', round(e0.elapsed_time(e1) * 1e3 / reps, 1))

need = lib.dcf_profile_report(None, 0)
buf = ctypes.create_string_buffer(int(need) + 16)
lib.dcf_profile_report(buf, len(buf))
for k, v in json.loads(buf.value.decode()).items():
    if 'gemm' in k: print('   kernel only:', k, round(1e3 * v['ms'] / v['count'], 1), 'us', round(v['flops'] / v['ms'] / 1e9, 1), 'TF')
