"""ctypes binding of ``librgrg_hip.so`` (C ABI declared in ``include/rgrg_hip.h``).

There is deliberately NO fallback: if the library is missing or fails to load the
product raises; if a call fails the HIP error string is raised.  torch only supplies
device memory (``tensor.data_ptr()``) and streams.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from . import build as _build

c_float_p = C.c_void_p  # device pointers travel as void*
_i, _f, _p = C.c_int, C.c_float, C.c_void_p

ACT_NONE, ACT_RELU, ACT_GELU_NEW = 0, 1, 2
ABI_VERSION = 20  # must equal rgrg_abi_version() of the loaded library; bump both on ANY signature change


class RgrgHipError(RuntimeError):
    pass


class DecoderLayerWeights(C.Structure):
    _fields_ = [(n, _p) for n in ("ln1_g", "ln1_b", "ln2_g", "ln2_b", "c_attn_w", "c_attn_b", "attn_proj_w",
                                  "attn_proj_b", "c_fc_w", "c_fc_b", "mlp_proj_w", "mlp_proj_b")]


class DecoderWeights(C.Structure):
    _fields_ = [("n_layer", _i), ("d_model", _i), ("n_head", _i), ("vocab", _i), ("wte", _p), ("lnf_g", _p),
                ("lnf_b", _p), ("fst0_w", _p), ("fst0_b", _p), ("fst2_w", _p), ("fst2_b", _p), ("ukv_w", _p),
                ("ukv_b", _p), ("layers", C.POINTER(DecoderLayerWeights))]


# name -> (restype, argtypes); mirrors include/rgrg_hip.h one to one
SIGNATURES = {
    "rgrg_last_error": (C.c_char_p, []),
    "rgrg_abi_version": (_i, []),
    "rgrg_device_arch": (_i, [_i, C.c_char_p, _i]),
    "rgrg_linear_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "rgrg_conv2d_nhwc_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "rgrg_stem_conv7x7_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "rgrg_maxpool3x3s2_nhwc_f32": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "rgrg_rpn_proposals_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _f, _f, _p]),
    "rgrg_roi_align_avgpool_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p]),
    "rgrg_roi_align_avgpool_bf16maps": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _i, _p]),
    "rgrg_top1_per_class_f32": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _f, _p]),
    "rgrg_select_regions_f32": (_i, [_p, _p, _f, _p, _p, _p, _i, _p]),
    "rgrg_bce_with_logits_masked_f32": (_i, [_p, _p, _p, C.c_float, _i, _p, _p]),
    "rgrg_preprocess_u8_f32": (_i, [_p, _i, _i, _i, _i, _i, C.c_float, C.c_float, _p, _p]),
    "rgrg_gather_rows_f32": (_i, [_p, _p, _p, _i, _i, _p]),
    "rgrg_decoder_create": (_i, [C.POINTER(DecoderWeights), _i, _i, C.POINTER(_p)]),
    "rgrg_decoder_create_with_cache": (_i, [C.POINTER(DecoderWeights), _i, _i, _p, C.c_size_t, C.POINTER(_p)]),
    "rgrg_decoder_kv_cache_bytes": (C.c_size_t, [_i, _i, _i]),
    "rgrg_decoder_destroy": (None, [_p]),
    "rgrg_decoder_generate": (_i, [_p, _p, _i, _i, _p, _i, C.POINTER(_i), _i, _p]),
    "rgrg_decoder_beam_search": (_i, [_p, _p, _i, _i, _i, _i, _f, _i, _p, _i, C.POINTER(_i), _p]),
    "rgrg_decoder_set_precision": (_i, [_p, _i]),
    "rgrg_decoder_lm_forward": (_i, [_p, _p, _p, _p, _i, _i, _p, _p, _p]),
    "rgrg_decoder_lm_loss_grad": (_i, [_p, _p, _p, _p, _i, _i, C.c_float, C.c_float, C.c_uint64, _p, _p, _p, _p, _p, _p, _p, _p]),
    "rgrg_dropout_mask_f32": (_i, [C.c_uint64, C.c_uint32, C.c_float, C.c_int64, _i, _p, _p]),
    "rgrg_decoder_refresh_trainable": (_i, [_p, _p]),
    "rgrg_decoder_take_id_error": (_i, [_p, C.POINTER(_i)]),
    "rgrg_decoder_forward_cached": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p, _p]),
    "rgrg_decoder_cache_plane": (_i, [_p, _i, _i, C.POINTER(C.c_void_p), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "rgrg_transpose_pad_f32": (_i, [_p, _p, _i, _i, _i, _p]),
    "rgrg_colsum_f32": (_i, [_p, _p, _i, _i, _p]),
    "rgrg_relu_backward_f32": (_i, [_p, _p, C.c_int64, _p]),
    "rgrg_bce_with_logits_masked_backward_f32": (_i, [_p, _p, _p, C.c_float, _i, C.c_float, _p, _i, _p]),
    "rgrg_adamw_step_f32": (_i, [_p, _p, _p, _p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _i, C.c_float, _p]),
    "rgrg_adamw_multi_step_f32": (_i, [_p, _i, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _i, C.c_float, _p]),
    "rgrg_f32_to_bf16": (_i, [_p, _p, C.c_int64, _i, _p]),
    "rgrg_bf16_to_f32": (_i, [_p, _p, C.c_int64, _i, _p]),
    "rgrg_conv2d_nhwc_bf16": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "rgrg_linear_bf16w_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "rgrg_linear_bf16_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "rgrg_debug_linear_bf16_tile": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "rgrg_debug_linear_bf16_train": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "rgrg_debug_linear_bf16_argmax": (_i, [_p, _p, _p, _i, _i, _i, _p, _p, _i, _p]),
    "rgrg_debug_ln_fold16": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "rgrg_debug_linear_bf16_ln": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "rgrg_debug_linear_bf16_ln_kp": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "rgrg_decoder_copy_last_logits": (_i, [_p, _p, _i, _p]),
    "rgrg_box_match_f32": (_i, [_p, _p, _i, _p, C.c_int64, _p, _i, _i, _f, _f, _i, _p, _p, _p]),
    "rgrg_balanced_sample": (_i, [_p, _p, _i, _p, _p, C.c_uint64, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p]),
    "rgrg_rpn_loss_sampled_f32": (_i, [_p, _i, _i, _p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _p, _p]),
    "rgrg_roi_add_gt_f32": (_i, [_p, _p, _i, _p, _p, _i, _i, _p, _p, _p]),
    "rgrg_roi_gather_samples_f32": (_i, [_p, _p, _p, _p, _p, _i, _p, _p, _i, _i, _i, _f, _f, _f, _f, _p, _p, _p, _p, _p]),
    "rgrg_fastrcnn_loss_f32": (_i, [_p, _i, _i, _p, _p, _i, _p, _p]),
    "rgrg_decoder_trace_step": (_i, [_p, _i, _i, _i, _p, _i, C.POINTER(_i)]),
    "rgrg_decoder_attention_only": (_i, [_p, _i, _i, _i, _p]),
    "rgrg_decoder_row_limit": (_i, [_p]),
    "rgrg_debug_hw_ids": (_i, [_p, _i, _p]),
    "rgrg_decoder_set_lm_positions": (_i, [_p, _p, C.c_int64]),
    "rgrg_decoder_time_step_parts": (_i, [_p, _i, _i, _i, _i, C.POINTER(_f), C.POINTER(_f), C.POINTER(C.c_double),
                                          C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i)]),
    "rgrg_decoder_time_train_gemms": (_i, [_p, _i, _i, _i, C.POINTER(_f), C.POINTER(C.c_double), C.POINTER(_i)]),
    "rgrg_debug_chain": (_i, [_i, _i, _i, C.POINTER(_f)]),
    "rgrg_debug_grid_barrier": (_i, [_i, _i, _i, C.POINTER(_f), C.POINTER(C.c_uint)]),
    "rgrg_debug_xcc_map": (_i, [_i, _i, C.POINTER(_i)]),
}

_lib: Optional[C.CDLL] = None


def library_path() -> str:
    return _build.LIB_PATH


def load() -> C.CDLL:
    """Load the in-tree shared library.  It is (re)built first when it is missing or older than its sources and hipcc
    is available; the ABI version of whatever gets loaded must match the ctypes signatures above (a stale library
    with new argument lists would corrupt memory silently).  Raises RgrgHipError when it cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if _build.is_stale():
        try:
            _build.build_library()
        except Exception as e:  # noqa: BLE001
            if not os.path.exists(path):
                raise RgrgHipError(f"librgrg_hip.so is missing and could not be built: {e}") from e
            # no toolchain on this box (e.g. a deployment image): the ABI check below decides
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise RgrgHipError(f"cannot load {path}: {e} (the RGRG HIP path has no CPU fallback)") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    got = lib.rgrg_abi_version()
    if got != ABI_VERSION:
        raise RgrgHipError(f"{path} has ABI version {got}, this package expects {ABI_VERSION}: rebuild it "
                           "(python -m rgrg_amd.build)")
    _lib = lib
    return lib


def autocast_mode() -> int:
    """The reduced-precision mode the caller asked for through torch.autocast("cuda", dtype): 0 = none (fp32), 1 = bfloat16,
    2 = float16 (the dtype of the reference's own scripts: generate_reports_for_images.py:108, train_full_model.py:172).  The HIP
    path then runs its matrix-core GEMMs / convolutions with 16-bit operands OF THAT TYPE and keeps the many-row K/V cache
    in it; everything else stays fp32."""
    import torch
    if not torch.is_autocast_enabled("cuda"):
        return 0
    dt = torch.get_autocast_dtype("cuda")
    return 1 if dt == torch.bfloat16 else (2 if dt == torch.float16 else 0)


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().rgrg_last_error()
        raise RgrgHipError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")


def ptr(t) -> Optional[int]:
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "rgrg_hip expects dense tensors"
    return t.data_ptr()
