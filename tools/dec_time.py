"""Attention half of a fusion layer as one kernel (csrc/dec_chain.hip) against the launches it replaces, inside dcf_op_decoder:
    python tools/dec_time.py [B] [T] [Lk]
prints the per-kernel times of the library's per-launch profile (dcf_profile_enable) with the kernel on and off.
DCF_PKG_ROOT: take the package from another root (tools/dc_stamp.sh: the stamped build; prints the in-kernel cycle shares)."""
import ctypes
import importlib
import json
import os
import sys

import torch

ROOT = os.environ.get('DCF_PKG_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('cvpr2025-decafnet_amd')
lib = pkg._lib.lib()
P = pkg._lib.ptr

SEG = ['load + stats', 'xq + exchange', 'barrier 2', 'conv + q_norm + split', 'stage wait vmcnt', 'stage barrier', 'Q_h MFMAs', 'q split',
       'S MFMAs', 'softmax + P split', 'O MFMAs + ctx split', 'proj MFMAs', 'proj epilogue', 'tail', '-', '-']


def shapes(E, TE):
    return {'ln_xattn_q.weight': (E, 1), 'ln_xattn_q.bias': (E, 1), 'ln_xattn_kv.weight': (TE, 1), 'ln_xattn_kv.bias': (TE, 1),
            'xattn.q_conv.conv.weight': (E, 1, 3), 'xattn.q_norm.weight': (E, 1), 'xattn.q_norm.bias': (E, 1),
            'xattn.xattn.query.weight': (E, E, 1), 'xattn.xattn.query.bias': (E,), 'xattn.xattn.key.weight': (E, TE, 1),
            'xattn.xattn.key.bias': (E,), 'xattn.xattn.value.weight': (E, TE, 1), 'xattn.xattn.value.bias': (E,),
            'xattn.xattn.proj.weight': (2 * E, E, 1), 'xattn.xattn.proj.bias': (2 * E,), 'ln_ffn.weight': (E, 1), 'ln_ffn.bias': (E, 1),
            'ffn.fc.weight': (4 * E, E, 1), 'ffn.fc.bias': (4 * E,), 'ffn.proj.weight': (E, 4 * E, 1), 'ffn.proj.bias': (E,),
            'drop_path_ffn.scale': (1, E, 1)}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    Lk = int(sys.argv[3]) if len(sys.argv) > 3 else 33
    E = TE = 256
    sd = pkg.synth.make_state_dict(shapes(E, TE), 7)
    keep = []
    c = pkg._lib.DcfConfig()
    for k, v in dict(D=32, E=E, TE=TE, vid_heads=4, fusion_heads=4, fusion_layers=0, n_embd_convs=0, n_stem=0, n_levels=1, win=9,
                     head_layers=0, sn=60, sratio=0.3, msf=1, norm=1, max_batch=8, gemm_mode=16).items():
        setattr(c, k, v)
    g = torch.Generator().manual_seed(3)
    X0 = torch.randn(B * T, E, generator=g).cuda()
    mask = torch.ones(B * T, dtype=torch.bool).cuda()
    texts = [torch.randn(TE, Lk, generator=g).cuda() for _ in range(B)]
    tp = (ctypes.c_void_p * B)(*[t.data_ptr() for t in texts])
    ln = (ctypes.c_int32 * B)(*[Lk] * B)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for chain_rows in (0, 1 << 30):
        pkg._lib.check(lib.dcf_debug_set_option(b'dec_chain_min_rows', chain_rows))
        for rep in range(3):
            h = ctypes.c_void_p()
            pkg._lib.check(lib.dcf_model_create(ctypes.byref(c), ctypes.byref(h)))
            for k, v in sd.items():
                t = v.contiguous().cuda()
                keep.append(t)
                shape = (ctypes.c_int64 * max(t.dim(), 1))(*(t.shape if t.dim() else (1,)))
                pkg._lib.check(lib.dcf_model_bind(h, f'd.{k}'.encode(), P(t), shape, max(t.dim(), 1)))
            X = X0.clone()
            lib.dcf_profile_enable(1)
            pkg._lib.check(lib.dcf_op_decoder(h, b'd', P(X), P(mask), B, T, tp, None, ln, st), 'dcf_op_decoder')
            torch.cuda.synchronize()
            need = lib.dcf_profile_report(None, 0)
            buf = ctypes.create_string_buffer(int(need) + 16)
            lib.dcf_profile_report(buf, len(buf))
            lib.dcf_profile_enable(0)
            lib.dcf_model_destroy(h)
        prof = json.loads(buf.value.decode())
        print(f'== B={B} T={T} Lk={Lk} chain={"on" if chain_rows == 0 else "off"}: {sum(v["ms"] for v in prof.values()) * 1e3:.1f} us in all')
        for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms']):
            print(f'   {k:40s} {v["ms"] * 1e3:9.1f} us  n={v["count"]}')
        if chain_rows == 0 and hasattr(lib, 'dcf_debug_dc_stamps'):
            out = (ctypes.c_ulonglong * 16)()
            lib.dcf_debug_dc_stamps.restype = ctypes.c_int
            if lib.dcf_debug_dc_stamps(out) == 0:
                tot = sum(out)
                print(f'   in-kernel stamps (wave 0 of workgroup 1): {tot} cycles')
                for n, v in zip(SEG, out):
                    if v:
                        print(f'      {n:28s} {v:9d}  {100.0 * v / tot:5.1f} %')
    pkg._lib.check(lib.dcf_debug_set_option(b'dec_chain_min_rows', -1))


if __name__ == '__main__':
    main()
