"""ctypes binding of libdecafnet_hip.so (the C ABI declared in include/decafnet_hip.h).

The product path has no CPU fallback: if the shared object is missing or cannot be loaded,
``lib()`` raises and every operator fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DCF_LIB_PATH: developer switch for A/B runs of two builds inside one GPU call (tools/); the product path is the in-tree build
SO_PATH = os.environ.get('DCF_LIB_PATH') or os.path.join(_HERE, 'libdecafnet_hip.so')
_LIB = None

c_f32p = ctypes.c_void_p      # device pointers are passed as raw addresses
c_u8p = ctypes.c_void_p
c_i32p = ctypes.c_void_p
c_i64p = ctypes.c_void_p
i32, i64, f32, f64 = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_double
vp = ctypes.c_void_p


class DcfConfig(ctypes.Structure):
    _fields_ = [(n, i32) for n in ('D', 'E', 'TE', 'vid_heads', 'fusion_heads', 'fusion_layers', 'n_embd_convs',
                                   'n_stem', 'n_levels', 'win', 'head_layers', 'sn')] + \
               [('sratio', f32)] + [(n, i32) for n in ('msf', 'norm', 'use_abs_pe', 'max_batch', 'gemm_mode', 'model_kind', 'second_fusion',
                                                      'text_in', 'text_layers', 'text_heads', 'text_abs_pe', 'text_bkgd',
                                                      'scat', 'sfonly', 'text_kind', 'xattn_affine', 'vid_stride', 'pool_only', 'attn_mode')]


# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header
SIGNATURES = {
    'dcf_last_error': (ctypes.c_char_p, []),
    'dcf_abi_version': (i32, []),
    'dcf_model_create': (i32, [ctypes.POINTER(DcfConfig), ctypes.POINTER(vp)]),
    'dcf_model_destroy': (None, [vp]),
    'dcf_model_bind': (i32, [vp, ctypes.c_char_p, c_f32p, ctypes.POINTER(i64), i32]),
    'dcf_model_set_pe': (i32, [vp, c_f32p, i64]),
    'dcf_model_set_text_pe': (i32, [vp, c_f32p, i64]),
    'dcf_text_encode': (i32, [vp, c_f32p, c_u8p, i32, c_f32p, c_u8p, vp]),
    'dcf_model_finalize': (i32, [vp, vp]),
    'dcf_numerics_status': (i32, [vp, i32, vp]),
    'dcf_numerics_status_async': (i32, [vp, vp, vp]),
    'dcf_points_per_query': (i64, [vp, i64]),
    'dcf_forward_eval': (i32, [vp, c_f32p, c_f32p, c_u8p, i64, i32, ctypes.POINTER(vp), ctypes.POINTER(vp),
                               ctypes.POINTER(i32), c_f32p, c_f32p, c_f32p, c_u8p, vp]),
    'dcf_forward_eval_videos': (i32, [vp, i32, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp), i64, ctypes.POINTER(i32),
                                      ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(i32), ctypes.POINTER(vp),
                                      c_f32p, c_f32p, c_u8p, vp]),
    'dcf_forward_train_videos': (i32, [vp, i32, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp), i64, ctypes.POINTER(i32),
                                       ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(i32), ctypes.POINTER(vp),
                                       c_f32p, c_f32p, c_f32p, c_u8p, vp]),
    'dcf_sigmoid_focal_loss': (i32, [c_f32p, c_f32p, c_u8p, i64, f32, f32, i32, c_f32p, c_f32p, c_i32p, vp]),
    'dcf_ctr_iou_loss': (i32, [c_f32p, c_f32p, c_u8p, i64, i32, f32, c_f32p, c_f32p, c_i32p, vp]),
    'dcf_forward_eval_gated': (i32, [vp, c_f32p, c_f32p, c_u8p, i64, i32, ctypes.POINTER(vp), ctypes.POINTER(vp),
                                     ctypes.POINTER(i32), c_f32p, c_f32p, c_f32p, c_u8p, vp]),
    'dcf_hybrid_phase1': (i32, [vp, i32, c_f32p, c_f32p, c_u8p, i64, i64, i32, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(i32),
                                c_f32p, c_f32p, vp]),
    'dcf_hybrid_phase2': (i32, [vp, c_f32p, c_u8p, i64, c_f32p, vp]),
    'dcf_hybrid_phase3': (i32, [vp, c_f32p, c_f32p, c_f32p, c_u8p, c_f32p, c_f32p, c_u8p, vp]),
    'dcf_debug_copy': (i32, [vp, i32, c_f32p, i64, vp]),
    'dcf_graph_active': (i32, [vp]),
    'dcf_debug_set_option': (i32, [ctypes.c_char_p, i32]),
    'dcf_calib_mfma_rate': (i32, [i32, i32, ctypes.POINTER(i32), ctypes.POINTER(f32)]),
    'dcf_model_set_graph_mode': (i32, [vp, i32]),
    'dcf_model_set_ln_carry': (i32, [vp, i32]),
    'dcf_profile_enable': (i32, [i32]),
    'dcf_profile_report': (i64, [ctypes.c_char_p, i64]),
    'dcf_collect_segments': (i32, [c_f32p, c_f32p, c_u8p, i32, i64, i32, f32, i32, f32, c_f32p, c_f32p, c_i32p, vp]),
    'dcf_collect_segments_ext': (i32, [c_f32p, c_f32p, c_u8p, c_f32p, i32, i64, i32, f32, i32, f32, c_f32p, c_f32p, c_i32p, vp]),
    'dcf_nms_1d': (i32, [c_f32p, c_f32p, c_i32p, i32, i32, i32, f32, c_i64p, c_i32p, vp]),
    'dcf_softnms_1d': (i32, [c_f32p, c_f32p, c_i32p, i32, i32, i32, f32, f32, f32, i32, i32, c_f32p, c_i64p, c_i32p, vp]),
    'dcf_segment_voting': (i32, [c_f32p, i32, c_i32p, i32, i32, c_f32p, c_f32p, c_i32p, i32, i32, f32, i32, c_f32p, vp]),
    'dcf_op_linear': (i32, [c_f32p, c_f32p, c_f32p, c_f32p, i32, i32, i32, i32, vp]),
    'dcf_op_linear_split': (i32, [c_f32p, c_f32p, c_f32p, c_f32p, i32, i32, i32, i32, i32, vp]),
    'dcf_op_linear_cm': (i32, [c_f32p, c_f32p, c_f32p, c_f32p, i32, i32, i32, vp]),
    'dcf_op_linear_ln': (i32, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, i32, i32, i32, i32, i32, vp]),
    'dcf_op_linear_ln_carry': (i32, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, i32, i32, i32, i32, i32, i32, vp]),
    'dcf_op_ffn': (i32, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_u8p, c_f32p, c_f32p, i32, i32, i32, vp]),
    'dcf_op_linear_cm_split': (i32, [c_f32p, c_f32p, c_f32p, c_f32p, i32, i32, i32, i32, vp]),
    'dcf_op_head': (i32, [c_f32p, c_u8p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, i32, i32, i32, i32, f32, i32, vp]),
    'dcf_op_conv3': (i32, [c_f32p, c_u8p, c_f32p, c_f32p, i32, i32, i32, i32, vp]),
    'dcf_op_conv3_split': (i32, [c_f32p, c_u8p, c_f32p, c_f32p, i32, i32, i32, i32, i32, vp]),
    'dcf_op_layernorm': (i32, [c_f32p, c_f32p, c_f32p, c_f32p, i32, i32, i32, vp]),
    'dcf_op_xattn': (i32, [c_f32p, c_f32p, c_f32p, c_u8p, c_f32p, i32, i32, i32, i32, i32, vp]),
    'dcf_op_local_attn': (i32, [c_f32p, c_f32p, c_f32p, c_u8p, c_f32p, i32, i32, i32, i32, i32, vp]),
    'dcf_op_sidekick': (i32, [c_f32p, c_f32p, c_f32p, i32, i32, i32, i32, vp]),
    'dcf_op_gate': (i32, [c_f32p, c_u8p, c_f32p, c_u8p, i32, i32, i32, f64, i32, vp]),
    'dcf_op_encoder': (i32, [vp, ctypes.c_char_p, c_f32p, c_u8p, i32, i32, i32, c_f32p, c_u8p, vp]),
    'dcf_op_enc_pre': (i32, [vp, ctypes.c_char_p, c_f32p, c_u8p, i32, i32, i32, c_f32p, c_f32p, c_f32p, c_f32p, vp]),
    'dcf_op_decoder': (i32, [vp, ctypes.c_char_p, c_f32p, c_u8p, i32, i32, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(i32), vp]),
    'dcf_op_tcn': (i32, [vp, ctypes.c_char_p, c_f32p, c_u8p, i32, i32, i32, i32, c_f32p, vp]),
}


def lib():
    """Load (once) and return the ctypes handle.  torch is imported first so that the HIP runtime
    already mapped by PyTorch-ROCm (same soname, libamdhip64.so.7) is the one this library binds to."""
    global _LIB
    if _LIB is None:
        import torch  # noqa: F401
        if not os.path.exists(SO_PATH):
            raise RuntimeError(f'{SO_PATH} is missing: build it with __graft_entry__.build() '
                               f'(there is no CPU fallback for the grounding path)')
        h = ctypes.CDLL(SO_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = h
    return _LIB


def check(rc, what=''):
    if rc != 0:
        msg = lib().dcf_last_error().decode(errors='replace')
        raise RuntimeError(f'decafnet_hip {what} failed: {msg}')


def ptr(t):
    """device/host address of a torch tensor (None -> NULL)"""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
