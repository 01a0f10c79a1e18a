"""ctypes binding of libanyref_hip.so (include/anyref_hip.h, include/anyref_hip_ops.h).

The shared library is built in-tree by `__graft_entry__.build()` / `make -C anyref_amd/csrc`.
There is no fallback: if it is missing, importing the compute path raises.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- BEFORE the library: torch brings its own HIP runtime, and a process that loads
# libanyref_hip.so (and with it /opt/rocm's runtime) first ends with "no HIP device visible" in anyref_create

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libanyref_hip.so")

ABI_VERSION = 2
F32, BF16, F16 = 0, 1, 2
MODE_PARITY, MODE_PERF, MODE_PERF_FP8W, MODE_PARITY16 = 0, 1, 2, 3


class AnyrefConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("mode", C.c_int32),
        ("clip_image", C.c_int32), ("clip_patch", C.c_int32), ("clip_dim", C.c_int32),
        ("clip_heads", C.c_int32), ("clip_layers_run", C.c_int32), ("clip_mlp", C.c_int32),
        ("clip_eps", C.c_float),
        ("llm_vocab", C.c_int32), ("llm_dim", C.c_int32), ("llm_heads", C.c_int32),
        ("llm_layers", C.c_int32), ("llm_mlp", C.c_int32), ("llm_max_seq", C.c_int32),
        ("llm_rms_eps", C.c_float), ("llm_rope_theta", C.c_float),
        ("sam_img", C.c_int32), ("sam_patch", C.c_int32), ("sam_dim", C.c_int32), ("sam_depth", C.c_int32),
        ("sam_heads", C.c_int32), ("sam_mlp_ratio", C.c_int32), ("sam_window", C.c_int32),
        ("sam_n_global", C.c_int32), ("sam_global_idx", C.c_int32 * 8),
        ("sam_out_chans", C.c_int32), ("dec_heads", C.c_int32), ("dec_mlp", C.c_int32),
        ("dec_depth", C.c_int32), ("num_mask_tokens", C.c_int32),
        ("out_dim", C.c_int32), ("audio_dim", C.c_int32),
        ("seg_lo", C.c_int32), ("seg_hi", C.c_int32),
        ("rephrase_weight", C.c_float),
        ("max_batch", C.c_int32), ("max_seg", C.c_int32),
        ("aud_dim", C.c_int32), ("aud_blocks", C.c_int32), ("aud_heads", C.c_int32), ("aud_mel", C.c_int32),
        ("aud_len", C.c_int32), ("aud_kernel", C.c_int32), ("aud_stride", C.c_int32), ("aud_clips", C.c_int32),
    ]


_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_L = C.c_int64

# name -> (restype, argtypes): every symbol the two headers declare
SYMBOLS = {
    "anyref_create": (_I, [C.POINTER(AnyrefConfig), _I, C.POINTER(_P)]),
    "anyref_destroy": (None, [_P]),
    "anyref_last_error": (C.c_char_p, [_P]),
    "anyref_set_weight": (_I, [_P, C.c_char_p, _P, _I, _I, C.POINTER(_L), _I]),
    "anyref_finalize": (_I, [_P]),
    "anyref_generate": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _P, _P, _I, _I, _P, _P, _P, _P, _L, _P,
                             _P, _P]),
    "anyref_forward_teacher": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _L, _P, _P,
                                    _P, _P]),
    "anyref_encode_images": (_I, [_P, _P, _P, _I, _P, _P]),
    "anyref_sam_encode": (_I, [_P, _P, _P, _I, _P]),
    "anyref_mask_decode": (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _P, _P]),
    "anyref_llm_forward": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _P, _P]),
    "anyref_project_audio": (_I, [_P, _P, _P, _I, _P]),
    "anyref_audio_encode": (_I, [_P, _P, _P, _I, _P]),
    "anyref_seg_tail": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _L, _P, _P]),
    "anyref_set_seg_range": (_I, [_P, _I, _I]),
    "anyref_set_overlap": (_I, [_P, _I]),
    "anyref_set_early_tail": (_I, [_P, _I]),
    "anyref_set_extra_event": (_I, [_P, _P]),
    "anyref_set_graphs": (_I, [_P, _I]),
    "anyref_set_side_share": (_I, [_P, _I, _I]),
    "anyref_profile_enable": (_I, [_P, _I]),
    "anyref_profile_config": (_I, [_P, C.c_char_p, _I]),
    "anyref_profile_collect": (_I, [_P]),
    "anyref_profile_calibrate": (_I, [_P, _P, C.POINTER(C.c_double)]),
    "anyref_profile_read": (_I, [_P, _I, C.c_char_p, _I, C.POINTER(C.c_double), C.POINTER(_L),
                                C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "anyref_stamps_enable": (_I, [_P, _I]),
    "anyref_stamps_collect": (_I, [_P, C.POINTER(_L)]),
    "anyref_stamps_dropped": (_I, [_P, C.POINTER(_L)]),
    "anyref_stamps_read": (_I, [_P, _L, C.c_char_p, _I, C.POINTER(C.c_double), C.POINTER(C.c_double),
                               C.POINTER(C.c_double), C.POINTER(_I)]),
    "anyref_stamps_spread": (_I, [_P, _L, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "anyref_device_bytes": (_L, [_P]),
    "anyref_mode_name": (C.c_char_p, [_P]),
    # kernel-level test entry points (anyref_hip_ops.h)
    "anyref_op_gemm": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I]),
    "anyref_op_gemm_fp8": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I]),
    "anyref_op_quant_fp8": (_I, [_P, _P, _I, _I, _P, _P]),
    "anyref_op_gemv_fp8": (_I, [_P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _I, _I, _I]),
    "anyref_op_iou_counts": (_I, [_P, _P, _P, _I, _L, _P]),
    "anyref_op_avs_counts": (_I, [_P, _P, _P, _I, _L, _P, _I, _F, _P, _P]),
    "anyref_op_sam_preprocess": (_I, [_P, _P, _I, _I, _I, C.POINTER(C.c_float), C.POINTER(C.c_float), _P]),
    "anyref_op_pil_resample_u8": (_I, [_P, _P, _I, _I, _I, _P, _P, _I, _I, _P, _P, _I, _P, _P, _I]),
    "anyref_op_pool_ref_tokens": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "anyref_op_kaldi_fbank": (_I, [_P, _P, _I, _I, _I, _I, _I, C.c_float, _P, _I, _P, _P, _I, C.c_float, C.c_float, _P]),
    "anyref_op_clip_finish": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, C.POINTER(C.c_float), C.POINTER(C.c_float), _P]),
    "anyref_op_gemv": (_I, [_I, _P, _P, _P, _F, _P, _P, _P, _P, _P, _I, _I, _I, _I]),
    "anyref_op_norm": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _F, _I]),
    "anyref_op_attention": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P, _P, _P, _I, _I]),
    "anyref_op_attention_tab": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _I, _I, _I]),
    "anyref_op_gemm_gather": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I]),
    "anyref_op_attention_relp": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _I, _I, _I]),
    "anyref_op_rel_pos": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "anyref_op_postprocess": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "anyref_op_last_error": (C.c_char_p, []),
}

_lib = None


def load() -> C.CDLL:
    """Load the HIP backend; raises (never falls back) when it is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build the HIP backend first "
            "(`python -c 'import __graft_entry__ as g; g.build()'` or `make -C anyref_amd/csrc`). "
            "anyref_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
