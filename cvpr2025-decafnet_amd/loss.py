"""Point losses of the training objective, forward values on the MI355X (csrc/loss.hip) with the reference's signatures
(libs/modeling/loss.py): ``sigmoid_focal_loss`` (:5-57), ``ctr_giou_loss`` (:60-109), ``ctr_diou_loss`` (:111-166), and the
two helpers the reference's Trainer wraps them in (libs/worker_v2.py:85-91).  No backward pass: the returned tensors carry no
autograd graph (training is out of scope).

The reference indexes with boolean masks before calling (``logits[fpn_masks]``, ``offsets[pos_masks]``,
worker_v2.py:446-458), which compacts on the device and synchronises the host for the output size; ``select=`` takes the mask
instead and leaves everything on the device.
"""
from __future__ import annotations

import torch

from . import _lib


def _flat_f32(x, name):
    if not x.is_cuda:
        raise RuntimeError(f'{name} must live on the MI355X: the losses have no CPU path')
    return x.float().contiguous()


def _reduce(elem, total, count, reduction, shape):
    if reduction == 'none':
        return elem.view(shape)
    if reduction == 'sum':
        return total[0]
    if reduction == 'mean':
        return total[0] / count[0].to(torch.float32)
    raise ValueError(f'reduction {reduction!r}')


def _selection(select, n, device):
    """the boolean mask the kernels index with: one byte per element, on the device of the operands (no broadcasting -- the
    kernel reads select[i] for every i < n)"""
    if select is None:
        return None
    if not select.is_cuda or select.device != device:
        raise ValueError(f'select must live on {device} (got {select.device})')
    if select.numel() != n:
        raise ValueError(f'select has {select.numel()} elements, the loss has {n}')
    return select.to(torch.bool).contiguous()


def sigmoid_focal_loss(inputs, targets, alpha: float = -1, gamma: float = 2.0, smoothing: bool = True, reduction: str = 'none',
                       select=None):
    """loss.py:5-57.  ``select`` (optional bool tensor of the same shape): only these elements count ('sum' / 'mean'); with
    reduction 'none' the unselected elements are 0."""
    x, t = _flat_f32(inputs, 'inputs'), _flat_f32(targets, 'targets')
    assert x.shape == t.shape
    n = x.numel()
    lib = _lib.lib()
    sel = _selection(select, n, x.device)
    elem = torch.empty_like(x) if reduction == 'none' else None
    total = torch.zeros(1, device=x.device) if reduction != 'none' else None
    count = torch.zeros(1, device=x.device, dtype=torch.int32) if reduction == 'mean' else None
    _lib.check(lib.dcf_sigmoid_focal_loss(_lib.ptr(x), _lib.ptr(t), _lib.ptr(sel), n, float(alpha), float(gamma), int(bool(smoothing)),
                                          _lib.ptr(elem), _lib.ptr(total), _lib.ptr(count), _lib.current_stream()), 'dcf_sigmoid_focal_loss')
    return _reduce(elem, total, count, reduction, inputs.shape)


def _ctr_iou(input_offsets, target_offsets, reduction, eps, kind, select):
    a, b = _flat_f32(input_offsets, 'input_offsets'), _flat_f32(target_offsets, 'target_offsets')
    assert a.shape == b.shape and a.shape[-1] == 2
    n = a.numel() // 2
    lib = _lib.lib()
    sel = _selection(select, n, a.device)
    elem = torch.empty(a.shape[:-1], device=a.device) if reduction == 'none' else None
    total = torch.zeros(1, device=a.device) if reduction != 'none' else None
    count = torch.zeros(1, device=a.device, dtype=torch.int32) if reduction == 'mean' else None
    _lib.check(lib.dcf_ctr_iou_loss(_lib.ptr(a), _lib.ptr(b), _lib.ptr(sel), n, kind, float(eps), _lib.ptr(elem), _lib.ptr(total),
                                    _lib.ptr(count), _lib.current_stream()), 'dcf_ctr_iou_loss')
    if reduction == 'mean':                                              # empty selection: `0.0 * loss.sum()` (loss.py:106,163)
        if n == 0:
            return torch.zeros((), device=a.device)
        return torch.where(count[0] > 0, total[0] / count[0].clamp(min=1).to(torch.float32), total.new_zeros(()))
    return _reduce(elem, total, count, reduction, a.shape[:-1])


def ctr_giou_loss(input_offsets, target_offsets, reduction: str = 'none', eps: float = 1e-8, select=None):
    """loss.py:60-109 (the generalised IoU reduces to the IoU for segments sharing a centre point).  The reference asserts
    non-negative offsets on the host (loss.py:87-88); RegHead ends with a ReLU (head.py:104), targets are distances."""
    return _ctr_iou(input_offsets, target_offsets, reduction, eps, 0, select)


def ctr_diou_loss(input_offsets, target_offsets, reduction: str = 'none', eps: float = 1e-8, select=None):
    """loss.py:111-166."""
    return _ctr_iou(input_offsets, target_offsets, reduction, eps, 1, select)


def calc_focal_loss(logits, labels, smoothing=0.2, alpha=0.5, reduction='sum', select=None):
    """worker_v2.py:85-87: label smoothing, then the focal loss."""
    labels = labels.to(logits.dtype) * (1.0 - smoothing) + smoothing / 2
    return sigmoid_focal_loss(logits, labels, alpha=alpha, reduction=reduction, select=select)


def calc_iou_loss(pred_offsets, gt_offsets, reg_loss='diou', reduction='sum', select=None):
    """worker_v2.py:89-91."""
    fn = ctr_diou_loss if reg_loss == 'diou' else ctr_giou_loss
    return fn(pred_offsets, gt_offsets, reduction=reduction, select=select)
