"""Plain-dict configuration tree for the grounding hot path.

The reference reads a yacs ``CfgNode`` (libs/core/opt.py:75-200).  yacs is not a
dependency of this package; ``AttrDict`` offers the subset of that interface the model
constructors use (attribute access, ``clone()``, ``deepcopy``, ``pop``, ``**`` expansion,
``get``), so the same object can be handed to the reference's constructors and to ours.
"""
from __future__ import annotations

import copy


class AttrDict(dict):
    """dict with attribute access and yacs-style ``clone()``."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def clone(self):
        return copy.deepcopy(self)


def to_attr(d):
    if isinstance(d, dict):
        return AttrDict({k: to_attr(v) for k, v in d.items()})
    if isinstance(d, list):
        return tuple(d)
    return d


def make_opt(D=1024, E=256, TE=256, text_in=300, n_levels=8, win=9, n_heads=4, sn=60, sratio=0.3,
             msf=True, scat=False, norm=True, max_seq_len=2304, text_layers=5, fusion_layers=2,
             text_max_len=48, n_embd_convs=2, n_stem=0, head_layers=2, use_abs_pe=True,
             text_use_abs_pe=False, max_vid_len=None, sfonly=False, text_name='transformer', text_bkgd=True, xattn_mode='adaln',
             vid_stride=1, pool_only=False):
    """Build an ``opt`` tree with the keys the hot path reads.  Defaults are
    the survey's probe configuration: D=1024, E=TE=256, L=8, w=9,
    4 heads, 2 fusion layers, sn=60, sratio=0.3, msf, norm."""
    opt = dict(
        model=dict(
            name='iter', sn=sn, sratio=sratio, msf=msf, scat=scat, sfonly=sfonly, norm=norm,
            max_vid_len=max_vid_len or max_seq_len, vid_stride=vid_stride,
            num_fpn_levels=n_levels, mha_win_size=win,
            text_net=(dict(name='transformer', in_dim=text_in, embd_dim=TE, max_seq_len=text_max_len,
                           n_heads=n_heads, n_layers=text_layers, attn_pdrop=0.0, proj_pdrop=0.0,
                           path_pdrop=0.0, use_abs_pe=text_use_abs_pe, use_bkgd_token=text_bkgd) if text_name == 'transformer' else
                      dict(name='identity', in_dim=text_in, embd_dim=TE, max_seq_len=text_max_len, n_heads=n_heads,
                           use_abs_pe=text_use_abs_pe, use_bkgd_token=text_bkgd)),
            vid_net=dict(name='transformer', in_dim=D, embd_dim=E, n_heads=n_heads,
                         max_seq_len=max_seq_len, stride=vid_stride, arch=(n_embd_convs, n_stem, n_levels),
                         mha_win_size=win, attn_pdrop=0.0, proj_pdrop=0.0, path_pdrop=0.0,
                         use_abs_pe=use_abs_pe, fuse='cat', pool_only=pool_only, cdrop=0.0),
            fusion=dict(name='xattn', vid_dim=E, text_dim=TE, n_layers=fusion_layers, n_heads=n_heads,
                        attn_pdrop=0.0, proj_pdrop=0.0, path_pdrop=0.0, xattn_mode=xattn_mode),
            cls_head=dict(name='cls', embd_dim=E, n_layers=head_layers, prior_prob=0.0),
            reg_head=dict(name='reg', embd_dim=E, num_fpn_levels=n_levels, n_layers=head_layers),
        ),
        pt_gen=dict(regression_range=4, sigma=0.5, num_fpn_levels=n_levels),
        eval=dict(ranks=(1, 5), iou_threshs=(0.3, 0.5), pre_nms_thresh=0.001, pre_nms_topk=2000,
                  seg_len_thresh=0.1),
        nms=dict(mode='soft_nms', iou_thresh=0.1, min_score=0.001, max_num_segs=5, sigma=0.9,
                 voting_thresh=0.95),
    )
    return to_attr(opt)
