"""ctypes binding of libsrganst.so (the C ABI declared in include/srganst.h).

The HIP library is the product: there is no CPU fallback.  If the shared object is missing
or a call fails, this module raises - loudly - instead of routing anywhere else.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SST_LIB_PATH") or os.path.join(_HERE, "lib", "libsrganst.so")      # SST_LIB_PATH: dev builds (ablation) of the library

P = c_void_p  # device pointer

# name -> (restype, argtypes).  Kept in sync with include/srganst.h (tests/test_abi_symbols.py).
SIGNATURES = {
    "sst_last_error": (c_char_p, []),
    "sst_version": (c_int, []),
    "sst_arch": (c_char_p, []),
    "sst_clear_error": (c_int, []),
    "sst_reload_env": (c_int, []),
    "sst_st_loss_workspace": (c_int, [c_int, c_int, c_int, POINTER(c_int64)]),
    "sst_st_loss_fwd": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_float, c_float, c_int, P]),
    "sst_st_loss_bwd": (c_int, [P, P, P, P, c_float, c_int, c_int, c_int, c_int, c_float, c_float, P]),
    "sst_st_pixel_loss_fwd": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_float, c_int, P]),
    "sst_st_pixel_loss_bwd": (c_int, [P, P, P, P, P, c_float, c_float, c_int, c_int, c_int, c_int, c_int, c_float, c_float, P]),
    "sst_conv_packed_floats": (c_int64, [c_int, c_int, c_int]),
    "sst_conv_pack": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "sst_conv_pack_multi": (c_int, [P, c_int, c_int, P]),
    "sst_conv_kernel_name": (c_char_p, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "sst_conv_wgrad_kernel_name": (c_char_p, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "sst_debug_big_tile_launches": (ctypes.c_long, []),
    "sst_adam_flat": (c_int, [P, P, P, P, c_int64, P, P, c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                            ctypes.c_double, P]),
    "sst_debug_band_launches": (ctypes.c_long, []),
    "sst_debug_wgrad_band_launches": (ctypes.c_long, []),
    "sst_debug_mfma_peak": (c_int, [P, c_int, c_int, P]),
    "sst_debug_bf16x3": (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    "sst_debug_stamp": (c_int, [P, c_int, P]),
    "sst_conv_mtiles": (c_int, [c_int, c_int, c_int]),
    "sst_conv_stat_tiles": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "sst_conv_fwd": (c_int, [P, P, P, P, P, P, P, P, c_float, c_int, P, P, P, c_int,
                             c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "sst_conv_pipe_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "sst_conv_pipe_stat_tiles": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "sst_conv_pipe_ws_floats": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "sst_conv_pipe_fwd": (c_int, [P, P, P, P, P, P, P, c_float, c_int, P, P, P, P, P, P, c_float, c_int, P, P,
                                  c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "sst_conv_pipe_groups_ok": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "sst_conv_pipe_fwd_grp": (c_int, [P, P, P, P, P, P, P, c_float, c_int, P, P, P, P, P, P, c_float, c_int, P, P,
                                      c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "sst_conv_ns_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "sst_conv_ns_fwd": (c_int, [P, P, P, P, P, P, P, c_float, c_int, P, P, P, P, P, P, c_float, c_int, P,
                                c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "sst_conv_s2_dgrad_pipe_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "sst_conv_s2_dgrad_pipe_ws_floats": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "sst_conv_s2_dgrad_pipe": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "sst_conv_s2_dgrad_pipe_stat_tiles": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "sst_conv_s2_dgrad_pipe_bwdstats": (c_int, [P, P, P, P, P, P, P, P, c_float, c_int, P, c_int, c_int, c_int, c_int, c_int, P]),
    "sst_conv_s2_dgrad_pipe_groups_ok": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "sst_conv_s2_dgrad_pipe_bwdstats_grp": (c_int, [P, P, P, P, P, P, P, P, c_float, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "sst_conv_acc_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "sst_conv_fwd_acc": (c_int, [P, P, P, P, P, P, P, P, c_float, c_int, P, P, P, c_float, c_float, c_float, P, P, P, P, P, P, P,
                                 c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "sst_conv_dgrad_fused_acc": (c_int, [P, P, P, P, P, c_float, c_int, P, P, P, P, P, P, P, P, c_float, c_int, P, P, P, P, c_float, P, P, P,
                                         P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "sst_bn_finalize_acc": (c_int, [P, c_int, c_int, c_float, P, P, P, P, P, P, P, P, c_float, c_float, P]),
    "sst_conv_fwd_resin": (c_int, [P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "sst_conv_dgrad_bwdstats": (c_int, [P, P, P, P, P, P, P, P, c_float, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "sst_conv_dgrad_fused": (c_int, [P, P, P, P, P, P, P, P, c_float, c_int, P, P, P, P, P, P, P, P, c_float, c_int, P,
                                     c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "sst_conv_wgrad_chunks": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "sst_conv_wgrad_chunks2": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "sst_conv_wgrad": (c_int, [P, P, P, P, P, P, P, c_float, c_int, c_int, c_int, c_int, c_int, c_int,
                               c_int, c_int, c_int, P]),
    "sst_conv_wgrad_groups_ok": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "sst_conv_wgrad_grp": (c_int, [P, P, P, P, P, P, P, c_float, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_int, c_int, c_int, c_int, P]),
    "sst_conv_wgrad_grouped": (c_int, [P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "sst_conv_wgrad_pending_reduce": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "sst_wgrad_reduce_multi": (c_int, [P, c_int, P]),
    "sst_bn_finalize": (c_int, [P, P, c_int, c_int, P, P, P, P, P, P, P, P, c_float, c_float, P]),
    "sst_bn_finalize_grp": (c_int, [P, P, c_int, c_int, c_int, P, P, P, P, P, P, P, P, c_float, c_float, P]),
    "sst_bn_eval_affine": (c_int, [P, P, P, P, P, P, c_int, c_float, P]),
    "sst_bn_residual": (c_int, [P, P, P, P, P, P, c_int64, c_int, P]),
    "sst_bwd_reduce_blocks": (c_int, [c_int64, c_int]),
    "sst_bwd_reduce": (c_int, [P, P, P, P, P, P, c_float, c_int, P, c_int64, c_int, P]),
    "sst_bwd_finalize": (c_int, [P, c_int, c_int, c_float, P, P, P, P, P, P, P, P, P, c_int, P]),
    "sst_bwd_reduce_grp": (c_int, [P, P, P, P, P, P, c_float, c_int, P, c_int64, c_int, c_int, P]),
    "sst_bwd_finalize_grp": (c_int, [P, c_int, c_int, c_float, c_int, P, P, P, P, P, P, P, P, c_int, P]),
    "sst_bwd_apply_grp": (c_int, [P, P, P, P, P, P, c_float, c_int, P, P, P, P, c_int64, c_int, c_int, P]),
    "sst_bwd_finalize_wide": (c_int, [P, c_int, c_int, c_float, P, P, P, P, P, P, P, P, P, c_int, P, P, P]),
    "sst_act_bwd_partial_blocks": (c_int, [c_int64]),
    "sst_act_bwd_partial": (c_int, [P, P, P, P, c_float, P, P, c_int64, c_int, c_int, c_int, P]),
    "sst_bwd_apply": (c_int, [P, P, P, P, P, P, c_float, c_int, P, P, P, P, c_int64, c_int, c_int, c_int, P]),
    "sst_add": (c_int, [P, P, P, c_int64, P]),
    "sst_transpose": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "sst_transpose_affine": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    "sst_maxpool_relu_fwd": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "sst_maxpool_relu_bwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    "sst_clamp_bwd_blocks": (c_int, [c_int, c_int, c_int]),
    "sst_clamp_bwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "sst_pixel_loss_blocks": (c_int, [c_int64]),
    "sst_pixel_loss_fwd": (c_int, [P, P, P, P, P, c_int64, c_int, P]),
    "sst_pixel_loss_bwd": (c_int, [P, P, P, P, c_float, c_int, c_int64, c_int, P]),
    "sst_bce_logits": (c_int, [P, c_float, P, P, P, c_float, c_int, P]),
    "sst_bicubic": (c_int, [P, P, P, P, P, P, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "sst_feat_loss_fwd": (c_int, [P, P, P, P, c_float, c_int, P, P, P, c_int64, c_int, P]),
    "sst_feat_loss_bwd": (c_int, [P, P, P, P, c_float, c_int, P, P, c_float, c_int, c_int64, c_int, P]),
    "sst_bb_blocks": (c_int, [c_int, c_int, c_int]),
    "sst_bb_feature_dim": (c_int, [c_int]),
    "sst_bb_patches": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "sst_bb_match": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_float, c_int, c_int, P, P]),
    "sst_bb_match_dist": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_float, c_int, c_int, P, c_int, P]),
    "sst_bbg_patches": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "sst_bbg_blocks": (c_int, [c_int, c_int]),
    "sst_bbg_unfold": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "sst_bbg_match": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_float, c_int, c_int, P]),
    "sst_bbg_fold": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "sst_weighted_sum": (c_int, [POINTER(c_void_p), POINTER(c_float), c_int, P, P, P]),
    "sst_wgrad_c3_supported": (c_int, [c_int, c_int]),
    "sst_wgrad_c3_slab_floats": (c_int64, [c_int, c_int, c_int, c_int]),
    "sst_wgrad_c3": (c_int, [P, P, P, P, P, c_float, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "sst_conv9_c3_packed_floats": (c_int64, [c_int]),
    "sst_conv9_c3_pack": (c_int, [P, P, c_int, c_int, c_int, P]),
    "sst_conv9_c3_fwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "sst_conv9_to3_packed_floats": (c_int64, [c_int]),
    "sst_conv9_to3_pack": (c_int, [P, P, c_int, P]),
    "sst_conv9_to3_fwd": (c_int, [P, P, P, P, P, P, c_float, c_int, c_int, c_int, c_int, c_int, P]),
    "sst_conv_s2_dgrad_tiles": (c_int, [c_int, c_int, c_int]),
    "sst_conv_s2_dgrad_kernel_name": (c_char_p, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "sst_conv_s2_dgrad_fused": (c_int, [P, P, P, P, P, P, P, P, c_float, c_int, P, P, P, P, P, P, P, c_float, c_int, P,
                                        c_int, c_int, c_int, c_int, c_int, P]),
    "sst_conv_s2_dgrad_packed_floats": (c_int64, [c_int, c_int]),
    "sst_conv_s2_dgrad_pack": (c_int, [P, P, c_int, c_int, P]),
    "sst_conv_s2_dgrad": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "sst_linear_ksplit": (c_int, [c_int, c_int, c_int]),
    "sst_linear_fwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, P]),
    "sst_linear_dgrad": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "sst_linear_wgrad": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "sst_head_fwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_float, P]),
    "sst_head_bwd": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_float, c_int, P]),
    "sst_flatten_act": (c_int, [P, P, P, c_float, c_int, P, c_int, c_int, c_int, P]),
    "sst_flatten_act_grp": (c_int, [P, P, P, c_float, c_int, P, c_int, c_int, c_int, c_int, P]),
}

_lib = None


class HipPathError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipPathError(
                f"{LIB_PATH} is missing - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C srgan-st_amd/csrc).  There is no CPU fallback for the HIP path.")
        # PyTorch-ROCm first: it brings its own libamdhip64, and the library must bind to THAT runtime (the one that owns the tensors'
        # device context and streams).  Loaded before torch, libsrganst.so pulls in the system ROCm runtime instead and its first
        # launch fails with "no ROCm-capable device is detected" (seen with build() followed by smoke() in one process).
        import torch  # noqa: F401
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)           # AttributeError if the .so is stale: also loud
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def reload_env() -> None:
    """The library reads each SST_* dev switch from the environment once; call this after changing one."""
    if _lib is not None:
        _lib.sst_reload_env()


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().sst_last_error().decode(errors="replace")
        raise HipPathError(f"{what} failed (code {rc}): {msg}")


def ptr(t) -> int:
    """Device pointer of a contiguous fp32/int32 CUDA(HIP) tensor; None -> NULL."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HipPathError("HIP path needs tensors on a ROCm device (no CPU fallback)")
    if not t.is_contiguous():
        raise HipPathError("HIP path needs contiguous tensors")
    return t.data_ptr()


def stream_ptr() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
