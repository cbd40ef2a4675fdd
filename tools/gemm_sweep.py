"""Sweep GEMM tile configurations on the shapes of the grounding forward (GPU only, dev tool).
usage: python tools/gemm_sweep.py  -> prints TFLOP/s per (shape, tile).  Each tile runs in its own process
because the override is read once (DCF_GEMM_CFG)."""
import ctypes, importlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [(16384, 256, 256), (16384, 1024, 256), (16384, 256, 1024), (16384, 512, 256), (8192, 256, 256), (8192, 1024, 256),
          (8192, 256, 1024), (4096, 256, 256), (4096, 1024, 256), (4096, 256, 1024), (32640, 256, 768), (49152, 256, 256), (49152, 1024, 256), (49152, 256, 1024), (81920, 256, 256), (81920, 1024, 256), (81920, 256, 1024)]
TILES = ['f16:auto', 'f16:64x256', 'f16:64x128', 'f16:128x256']

def child():
    import torch
    pkg = importlib.import_module('cvpr2025-decafnet_amd')
    lib = pkg._lib.lib()
    out = {}
    for (M, N, K) in SHAPES:
        A = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda'); b = torch.randn(N, device='cuda')
        C = torch.empty(M, N, device='cuda')
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        mode = os.environ.get('SWEEP_MODE', 'f32')
        call = (lambda: lib.dcf_op_linear(P(A), P(W), P(b), P(C), M, N, K, 0, st)) if mode == 'f32' else \
               (lambda: lib.dcf_op_linear_split(P(A), P(W), P(b), P(C), M, N, K, 0, 6 if mode == 'x6' else 16, st))
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        # kernel-only time from the library's per-launch profiler (the x6/x3 operator also splits W on every call)
        lib.dcf_profile_enable(1)
        reps = 20
        for _ in range(reps):
            call()
        torch.cuda.synchronize()
        need = lib.dcf_profile_report(None, 0)
        buf = ctypes.create_string_buffer(int(need) + 16)
        lib.dcf_profile_report(buf, len(buf))
        lib.dcf_profile_enable(0)
        prof = json.loads(buf.value.decode())
        us = sum(1e3 * v['ms'] / v['count'] for k, v in prof.items() if k.startswith('gemm'))
        out[f'{M}x{N}x{K}'] = (us, 2.0 * M * N * K / us / 1e6)
    print(json.dumps(out))

if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        child()
    else:
        res = {}
        for t in TILES:
            env = dict(os.environ)
            mode, tile = t.split(':')
            env['SWEEP_MODE'] = mode
            if tile != 'auto':
                env['DCF_GEMM_CFG'] = tile
            r = subprocess.run([sys.executable, __file__, 'child'], env=env, capture_output=True, text=True)
            try:
                res[t] = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception:
                print(t, 'failed', r.stderr[-500:])
        print('%-18s' % 'shape MxNxK' + ''.join('%16s' % t for t in res))
        for (M, N, K) in SHAPES:
            k = f'{M}x{N}x{K}'
            print('%-18s' % k + ''.join('%9.1fus %4.0fTF' % tuple(res[t][k]) for t in res))
