"""Drop-in for the reference's compiled extension module ``nms_1d_cpu_vg``
(libs/nms/src/nms_cpu.cpp:184-194, built by libs/nms/setup_nms.py): same module name, same
two functions, same keyword names -- executed by the HIP kernels of libdecafnet_hip.so.

    import nms_1d_cpu_vg
    idx = nms_1d_cpu_vg.nms(segs, scores, iou_thresh=0.5)
    idx = nms_1d_cpu_vg.softnms(segs, scores, dets, iou_thresh=0.1, sigma=0.9, min_score=0.001, method=2)
"""
import importlib as _importlib

_impl = _importlib.import_module('cvpr2025-decafnet_amd.nms')


def nms(segs, scores, iou_thresh):
    """nms (HIP): greedy 1-D NMS, returns kept indices (int64) in descending score order."""
    return _impl.nms(segs, scores, iou_thresh)


def softnms(segs, scores, dets, iou_thresh, sigma, min_score, method):
    """softnms (HIP): method 0 vanilla / 1 linear / 2 gaussian; fills ``dets`` in place."""
    return _impl.softnms(segs, scores, dets, iou_thresh, sigma, min_score, method)
