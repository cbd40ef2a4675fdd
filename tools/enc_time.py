"""Encoder layer with the chain kernels (csrc/enc_chain.hip) against the launches they replace, inside dcf_op_encoder:
    python tools/enc_time.py [B] [T] [stride]
prints the per-kernel times of the library's per-launch profile (dcf_profile_enable) with the kernels on and off.
DCF_PKG_ROOT: take the package from another root (ablation builds)."""
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


EA_SEG = ['setup + head 0 requests', 'halo + next head requests', 'Q / K / halo split', 'V split', 'S MFMAs', 'band + softmax', 'P split',
          'O MFMAs + ctx split', 'projection MFMAs + epilogue', 'stage wait + barrier', 'last wait', 'last epilogue + stats']


def shapes(E):
    sh = {'ln_attn.weight': (E, 1), 'ln_attn.bias': (E, 1), 'ln_ffn.weight': (E, 1), 'ln_ffn.bias': (E, 1),
          'drop_path_attn.scale': (1, E, 1), 'drop_path_ffn.scale': (1, E, 1),
          'ffn.fc.weight': (4 * E, E, 1), 'ffn.fc.bias': (4 * E,), 'ffn.proj.weight': (E, 4 * E, 1), 'ffn.proj.bias': (E,)}
    for n in 'qkv':
        sh[f'attn.{n}_conv.conv.weight'] = (E, 1, 3)
        sh[f'attn.{n}_norm.weight'] = (E, 1)
        sh[f'attn.{n}_norm.bias'] = (E, 1)
    for n in ('query', 'key', 'value', 'proj'):
        sh[f'attn.attn.{n}.weight'] = (E, E, 1)
        sh[f'attn.attn.{n}.bias'] = (E,)
    return sh


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    stride = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    E = 256
    sd = pkg.synth.make_state_dict(shapes(E), 7)
    keep = []
    c = pkg._lib.DcfConfig()
    for k, v in dict(D=32, E=E, TE=E, vid_heads=4, fusion_heads=4, fusion_layers=0, n_embd_convs=0, n_stem=0, n_levels=1, win=9,
                     head_layers=0, sn=60, sratio=0.3, msf=1, norm=1, max_batch=8, gemm_mode=16).items():
        setattr(c, k, v)
    g = torch.Generator().manual_seed(3)
    X = torch.randn(B * T, E, generator=g).cuda()
    mask = torch.ones(B * T, dtype=torch.bool).cuda()
    To = T // stride
    Y = torch.empty(B * To, E, device='cuda')
    mo = torch.empty(B * To, dtype=torch.bool, device='cuda')
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for chain_rows in (0, 1 << 30):
        for o in (b'enc_chain_min_rows', b'enc_attn_min_rows'):
            pkg._lib.check(lib.dcf_debug_set_option(o, chain_rows))
        for rep in range(3):
            h = ctypes.c_void_p()
            pkg._lib.check(lib.dcf_model_create(ctypes.byref(c), ctypes.byref(h)))
            for k, v in sd.items():
                t = v.contiguous().cuda()
                keep.append(t)
                shape = (ctypes.c_int64 * max(t.dim(), 1))(*(t.shape if t.dim() else (1,)))
                pkg._lib.check(lib.dcf_model_bind(h, f'e.{k}'.encode(), P(t), shape, max(t.dim(), 1)))
            lib.dcf_profile_enable(1)
            pkg._lib.check(lib.dcf_op_encoder(h, b'e', P(X), P(mask), B, T, stride, P(Y), P(mo), st), 'dcf_op_encoder')
            torch.cuda.synchronize()
            need = lib.dcf_profile_report(None, 0)
            buf = ctypes.create_string_buffer(int(need) + 16)
            lib.dcf_profile_report(buf, len(buf))
            lib.dcf_profile_enable(0)
            lib.dcf_model_destroy(h)
        prof = json.loads(buf.value.decode())
        print(f'== B={B} T={T} stride={stride} chain={"on" if chain_rows == 0 else "off"}: {sum(v["ms"] for v in prof.values()) * 1e3:.1f} us in all')
        for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms']):
            print(f'   {k:40s} {v["ms"] * 1e3:9.1f} us  n={v["count"]}')
        if chain_rows == 0 and hasattr(lib, 'dcf_debug_ea_stamps'):      # diagnostic build of tools/ea_stamp.sh
            out = (ctypes.c_ulonglong * 16)()
            lib.dcf_debug_ea_stamps.restype = ctypes.c_int
            if lib.dcf_debug_ea_stamps(out) == 0:
                tot = sum(out)
                print(f'   k_enc_attn in-kernel stamps (wave 0 of workgroup 1): {tot} cycles')
                for n, v in zip(EA_SEG, out):
                    if v:
                        print(f'      {n:28s} {v:9d}  {100.0 * v / tot:5.1f} %')
    for o in (b'enc_chain_min_rows', b'enc_attn_min_rows'):
        pkg._lib.check(lib.dcf_debug_set_option(o, -1))


if __name__ == '__main__':
    main()
