#!/usr/bin/env python3
"""One-sided receptive field of the grounding forward, PROBED on the CPU oracle (float64): perturb one clip of a long
video (gate fixed, so the only global dependency is switched off) and find the farthest output position, at any pyramid
level, that changes.  dist.receptive_field() is the closed form; this tool is what it was validated against.

    python tools/receptive_field.py [L] [win] [fusion_layers] [n_embd_convs] [head_layers]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib  # noqa: E402
from oracle import decafnet_ref as R  # noqa: E402

pkg = importlib.import_module('cvpr2025-decafnet_amd')


@torch.no_grad()
def probe(L=8, win=9, fusion_layers=2, n_embd_convs=2, head_layers=2, E=32, D=32):
    torch.set_default_dtype(torch.float64)
    a = (2 ** (L - 1)) * max(win // 2, 1)
    rf_guess = pkg.dist.receptive_field(L, win, fusion_layers=fusion_layers, n_embd_convs=n_embd_convs, n_stem=0, head_layers=head_layers)
    T = -(-(2 * rf_guess + 4 * a) // a) * a
    kw = dict(D=D, E=E, TE=32, text_in=32, n_levels=L, win=win, n_heads=4, sn=60, sratio=0.3, msf=True, norm=True,
              max_seq_len=T, text_layers=1, text_max_len=24, fusion_layers=fusion_layers, n_embd_convs=n_embd_convs,
              use_abs_pe=False)
    opt = pkg.config.make_opt(**kw)
    opt.model['cls_head']['n_layers'] = head_layers
    opt.model['reg_head']['n_layers'] = head_layers
    model = pkg.modeling.create_model(opt)
    shapes = {k: list(v.shape) for k, v in model.state_dict().items()}
    sd = {k: v.double() for k, v in pkg.synth.make_state_dict(shapes, 3).items()}
    inp = pkg.synth.make_inputs(D, T, T, 1, 32, 8, 4)
    vid, sh = inp['vid'][0].double(), inp['shallow_vid'][0].double()
    mask = torch.ones(1, T, dtype=torch.bool)
    gate = torch.ones(1, T)
    t, m = R.encode_text(sd, opt.model, inp['tokens'][0][None].double(), torch.ones(1, 1, 8, dtype=torch.bool))
    base = R.forward_eval_window(sd, opt.model, vid[None], sh[None], mask, [t], [m], gate)
    out = {}
    for t0 in (T // 2, T // 2 + 2 ** (L - 1) - 1, T // 2 + 1):        # different phases of the stride-2 grid
        v2, s2 = vid.clone(), sh.clone()
        v2[:, t0] += 3.0
        s2[:, t0] -= 2.0
        got = R.forward_eval_window(sd, opt.model, v2[None], s2[None], mask, [t], [m], gate)
        left = right = 0
        for part in (0, 1):
            for l in range(L):
                d = (got[part][0][l] - base[part][0][l]).abs()
                d = d.reshape(d.shape[1], -1).amax(-1) if d.dim() == 3 else d[0]
                idx = torch.nonzero(d > 0).flatten()
                if len(idx):
                    lo, hi = int(idx[0]) << l, int(idx[-1]) << l
                    left, right = max(left, t0 - lo), max(right, hi - t0)
        out[t0] = (left, right)
    torch.set_default_dtype(torch.float32)
    return out, rf_guess


if __name__ == '__main__':
    args = [int(x) for x in sys.argv[1:]]
    res, formula = probe(*args)
    # an output at position p depends on inputs in [p - reach_left_of_output, p + reach_right_of_output]; perturbing input t0
    # changes outputs up to `right` clips to its right (they reach LEFT that far) and `left` clips to its left
    print({'probe (outputs changed left of / right of the perturbed clip)': res, 'dist.receptive_field': formula})
