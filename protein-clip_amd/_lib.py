"""ctypes binding of libpclip.so (the C ABI declared in include/pclip.h).

There is no CPU fallback: if the shared library is missing, or a call is rejected, this
module raises.  Tensors are passed as raw device pointers together with the current HIP
stream of the tensor's device; nothing here allocates device memory except through torch.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpclip.so")

_P = c_void_p

_SIGNATURES = {
    "pclip_version": (c_int, []),
    "pclip_last_error": (c_char_p, []),
    "pclip_check_device": (c_int, [c_int]),
    "pclip_l2norm_scale_fwd": (c_int, [_P, _P, _P, c_int, c_int, _P]),
    "pclip_l2norm_scale_bwd": (c_int, [_P, _P, _P, c_int, c_size_t, _P, _P, _P, c_int, c_int, _P]),
    "pclip_clip_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "pclip_clip_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "pclip_clip_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "pclip_clip_step_workspace_bytes": (c_size_t, [c_int, c_int]),
    "pclip_clip_step": (c_int, [_P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, c_size_t, _P]),
    "pclip_clip_loss_grad_workspace_bytes": (c_size_t, [c_int, c_int]),
    "pclip_clip_loss_grad": (c_int, [_P, _P, c_int, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "pclip_clip_step_sync_status": (c_int, [_P]),
    "pclip_clip_step_sync_reset": (c_int, [_P]),
    "pclip_clip_step_selftest_giveup": (c_int, [c_int]),
    "pclip_clip_step_noalloc_max_bytes": (c_size_t, []),
    "pclip_clip_step_noalloc": (c_int, [_P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P]),
    "pclip_rowce_fwd": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "pclip_rowce_bwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, c_float, _P, _P]),
    "pclip_filip_loss": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "pclip_masked_mean_fwd": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P]),
    "pclip_masked_mean_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, _P]),
    "pclip_head_tail_save_floats": (c_size_t, [c_int, c_int, c_int]),
    "pclip_head_tail_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "pclip_head_tail_fwd": (c_int, [_P, _P, c_int, _P, c_float, c_float, _P, c_int, c_int, _P, _P, _P]),
    "pclip_head_tail_bwd": (c_int, [_P, _P, c_int, _P, c_float, _P, _P, _P, c_int, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "pclip_filip_prep": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "pclip_filip_sim_fwd": (c_int, [_P, _P, c_int, c_int, _P, _P, c_int, c_int, _P, c_int, _P, _P, _P, _P, _P]),
    "pclip_filip_sim_fwd_cols": (c_int, [_P, _P, c_int, c_int, _P, _P, c_int, c_int, _P, c_int, _P, _P, _P, _P, c_int, c_int, _P]),
    "pclip_filip_sim_fwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "pclip_filip_sim_fwd_ws": (c_int, [_P, _P, c_int, c_int, _P, _P, c_int, c_int, _P, c_int, _P, _P, _P, _P, c_int, c_int, _P, c_size_t, _P]),
    "pclip_filip_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "pclip_filip_sim_bwd": (c_int, [_P, _P, c_int, c_int, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "pclip_filip_sim_bwd_resume": (c_int, [_P, _P, c_int, c_int, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "pclip_filip_bwd_fused_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "pclip_filip_bwd_fused": (c_int, [_P, _P, c_int, c_int, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "pclip_filip_prep_bwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, c_int, _P, _P]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


class PclipError(RuntimeError):
    pass


def load():
    """Load libpclip.so once and attach argument/return types.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PclipError(
            f"{LIB_PATH} not found: build it with `make -C protein-clip_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback for the HIP path"
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())


def stream_of(t: torch.Tensor):
    return c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def check(rc: int, what: str):
    if rc != 0:
        msg = load().pclip_last_error().decode(errors="replace")
        raise PclipError(f"{what} failed ({rc}): {msg}")


def require_gpu_tensor(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise PclipError(f"{name}: expected a GPU tensor (the HIP path has no CPU fallback), got device {t.device}")
    if t.dtype != dtype:
        raise PclipError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise PclipError(f"{name}: expected a contiguous tensor")
