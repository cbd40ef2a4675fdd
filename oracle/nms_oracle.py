"""ctypes front-end of oracle/nms_ref.c (CPU ORACLE -- test infrastructure only).

Exposes ``nms`` / ``softnms`` with the calling convention of the reference extension
``nms_1d_cpu_vg`` (libs/nms/src/nms_cpu.cpp:184-194) so tests can swap the oracle, the
compiled reference (oracle/_ref) and the HIP product path freely.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    so = os.path.join(_HERE, 'libnms_oracle.so')
    src = os.path.join(_HERE, 'nms_ref.c')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, 'libnms_oracle.so'], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        f32p = ctypes.POINTER(ctypes.c_float)
        i64p = ctypes.POINTER(ctypes.c_int64)
        _LIB.dcf_oracle_nms_1d.restype = ctypes.c_int64
        _LIB.dcf_oracle_nms_1d.argtypes = [f32p, f32p, ctypes.c_int64, ctypes.c_float, i64p]
        _LIB.dcf_oracle_softnms_1d.restype = ctypes.c_int64
        _LIB.dcf_oracle_softnms_1d.argtypes = [f32p, f32p, ctypes.c_int64, f32p, ctypes.c_float,
                                                ctypes.c_float, ctypes.c_float, ctypes.c_int, i64p]
    return _LIB


def _f32(t):
    a = np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def nms(segs, scores, iou_thresh):
    n = int(segs.shape[0])
    if segs.numel() == 0:
        return torch.empty(0, dtype=torch.int64)
    s, sp = _f32(segs)
    c, cp = _f32(scores)
    keep = np.empty(n, dtype=np.int64)
    k = lib().dcf_oracle_nms_1d(sp, cp, n, float(iou_thresh), keep.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    return torch.from_numpy(keep[:k].copy())


def softnms(segs, scores, dets, iou_thresh, sigma, min_score, method):
    n = int(segs.shape[0])
    if segs.numel() == 0:
        return torch.empty(0, dtype=torch.int64)
    s, sp = _f32(segs)
    c, cp = _f32(scores)
    assert dets.dtype == torch.float32 and dets.is_contiguous() and tuple(dets.shape) == (n, 3)
    d = dets.numpy()
    inds = np.empty(n, dtype=np.int64)
    k = lib().dcf_oracle_softnms_1d(sp, cp, n, d.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                    float(iou_thresh), float(sigma), float(min_score), int(method),
                                    inds.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    return torch.from_numpy(inds[:k].copy())
