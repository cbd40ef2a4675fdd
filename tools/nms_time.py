"""Kernel time of dcf_nms_1d (profiler label nms_1d) for a few (n, iou_thresh): python tools/nms_time.py"""
import sys, os, ctypes, json, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module('cvpr2025-decafnet_amd'); lib = pkg._lib.lib()
g = torch.Generator().manual_seed(0)
for n, thr in [(2000, 0.5), (2000, 2.0), (2000, 0.01), (64, 0.5), (512, 0.5), (1024, 0.5)]:
    c = torch.rand(n, generator=g) * 16000; w = torch.rand(n, generator=g) * 400 + 5
    segs = torch.stack([c - w / 2, c + w / 2], 1)[None].cuda().contiguous(); sc = torch.rand(1, n, generator=g).cuda()
    for _ in range(3): pkg.nms.nms_device(segs, sc, None, n, n, thr)
    torch.cuda.synchronize()
    lib.dcf_profile_enable(1)
    for _ in range(10): keep, kc = pkg.nms.nms_device(segs, sc, None, n, n, thr)
    torch.cuda.synchronize()
    need = lib.dcf_profile_report(None, 0); buf = ctypes.create_string_buffer(int(need) + 16); lib.dcf_profile_report(buf, len(buf))
    lib.dcf_profile_enable(0)
    print(n, thr, {k: round(1e3 * v['ms'] / v['count'], 1) for k, v in json.loads(buf.value.decode()).items()}, 'kept', int(kc))
