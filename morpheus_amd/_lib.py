"""ctypes binding of the C ABI declared in include/morpheus_hip.h.

The product path has no CPU fallback: if the library is missing or a call fails this raises.
"""
from __future__ import annotations

import ctypes
import os

import torch

from .build import SO as _BUILT_SO

# MORPHEUS_HIP_LIB: load another build of the SAME library (an A/B variant compiled with extra -D flags, tools/gpu/*.sh);
# the default is the in-tree build, and there is still no fallback to anything that is not this library
SO = os.environ.get("MORPHEUS_HIP_LIB") or _BUILT_SO

_P = ctypes.c_void_p
_I32, _I64, _F = ctypes.c_int32, ctypes.c_int64, ctypes.c_float

_SIGS = {
    "mh_abi_version": (ctypes.c_int, []),
    "mh_status_string": (ctypes.c_char_p, [ctypes.c_int]),
    "mh_grid_encode_fwd": (ctypes.c_int, [_P, _P, _P, _P, _P, _I64, _I32, _I32, _F, _I32, _P]),
    "mh_grid_encode_bwd": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _F, _P]),
    "mh_grid_bin_workspace_ints": (_I64, []),
    "mh_grid_bin_bricks": (_I32, []),
    "mh_grid_bin_index_ints": (_I32, []),
    "mh_grid_stage_min_points": (ctypes.c_int64, [ctypes.c_int64]),
    "mh_grid_bin_points": (ctypes.c_int, [_P, _I64, _F, _P, _P, _P, _P]),
    "mh_grid_encode_fwd_binned": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _F, _P]),
    "mh_grid_encode_bwd_binned": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I64, _I32, _I32, _F, _P, _P]),
    "mh_composite_fwd": (ctypes.c_int, [_P] * 10 + [_I32, _P]),
    "mh_composite_bwd": (ctypes.c_int, [_P] * 13 + [_I32, _P]),
    "mh_generate_rays": (ctypes.c_int, [_F, _F, _F, _F, _P, _I32, _I32, _P, _P, _P]),
    "mh_sample_uniform": (ctypes.c_int, [_P, _P, _P, _I32, _I32, _F, _P, _P, _P, _P, _P, _P, _P]),
    "mh_rays_sample_uniform": (ctypes.c_int, [_F, _F, _F, _F, _P, _I32, _I32, _P, _P, _I32, _I32, _F, _P, _P, _P, _P,
                                              _P, _P, _P, _P, _P]),
    "mh_march_cap": (_I32, [_F, _F]),
    "mh_march_slots": (ctypes.c_int, [_P, _P, _P, _I32, _F, _F, _I32, _P, _I32, _P, _P, _P, _P, _P]),
    "mh_march_pack": (ctypes.c_int, [_P, _P, _P, _P, _I32, _I32, _P, _P, _P, _P]),
    "mh_fd_taps": (ctypes.c_int, [_P, _P, _I32, _F, _F, _I64, _P, _P, _P]),
    "mh_fd_taps_bwd": (ctypes.c_int, [_P, _P, _P, _I32, _F, _F, _I64, _P, _P, _P]),
    "mh_fd_normal_fwd": (ctypes.c_int, [_P, _F, _I64, _P, _P, _P]),
    "mh_fd_normal_bwd": (ctypes.c_int, [_P, _P, _P, _F, _I64, _P, _P]),
    "mh_multicode_fwd": (ctypes.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P]),
    "mh_multicode_bwd": (ctypes.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "mh_sdf_losses_fwd": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _F, _I64, _P, _P, _P]),
    "mh_sdf_losses_bwd": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _F, _I64, _P, _P, _P, _P, _P, _P]),
    "mh_sample_positions": (ctypes.c_int, [_P, _P, _P, _P, _P, _I64, _P, _P]),
    "mh_sample_positions_bwd": (ctypes.c_int, [_P, _P, _P, _P, _P, _I32, _P, _P, _P]),
    "mh_mlp_tiles": (_I64, [_I64]),
    "mh_warp_acts_floats": (_I64, [_I64]),
    "mh_warp_dpre_floats": (_I64, [_I64]),
    "mh_warp_wpack_floats": (_I64, []),
    "mh_warp_wpackT_floats": (_I64, []),
    "mh_field_acts_floats": (_I64, [_I64]),
    "mh_field_wpack_floats": (_I64, []),
    "mh_field_wpackT_floats": (_I64, []),
    "mh_warp_fwd": (ctypes.c_int, [_P] * 8 + [_I32, _P, _P, _P, _I64, _P]),
    "mh_warp_bwd_data": (ctypes.c_int, [_P] * 5 + [_I32, _P, _P, _P, _I64, _P]),
    "mh_b3_slice": (ctypes.c_int, [_P, _P, _I32, _P, _P, _P, _P]),
    "mh_warp_w3_bytes": (_I64, []),
    "mh_field_w3_bytes": (_I64, []),
    "mh_field_fwd_b3": (ctypes.c_int, [_P] * 7 + [_I32, _I32, _P, _P, _P, _P, _I64, _P]),
    "mh_warp_w3T_bytes": (_I64, []),
    "mh_warp_bwd_data_b3": (ctypes.c_int, [_P] * 5 + [_I32, _P, _P, _P, _I64, _I32, _P]),
    "mh_warp_regen_dpre4": (_I32, [_I64]),
    "mh_warp_wgrad_workspace_floats": (_I64, [_I64]),
    "mh_warp_wgrad_b3": (ctypes.c_int, [_P] * 6 + [_I32, _P, _P, _P, _I64, _P]),
    "mh_warp_fwd_b3": (ctypes.c_int, [_P] * 8 + [_I32, _P, _P, _P, _I64, _P]),
    "mh_field_fwd": (ctypes.c_int, [_P] * 7 + [_I32, _I32, _P, _P, _P, _P, _I64, _P]),
    "mh_field_bwd_fused_workspace_floats": (_I64, [_I64]),
    "mh_field_dgeo_floats": (_I64, [_I64]),
    "mh_field_bwd_fused": (ctypes.c_int, [_P] * 8 + [_I32, _I32] + [_P] * 4 + [_I32] + [_P] * 5 + [_I64, _P]),
    "mh_field_bwd_fused_b3": (ctypes.c_int, [_P] * 8 + [_I32, _I32] + [_P] * 4 + [_I32] + [_P] * 5 + [_I64, _P]),
    "mh_field_w3T_bytes": (_I64, []),
    "mh_mlp_wgrad_workspace_floats": (_I64, [_I32, _P, _P, _I64]),
    "mh_mlp_wgrad": (ctypes.c_int, [_P, _P, _I64, _I64, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P]),
    "mh_mlp_wgrad_b3": (ctypes.c_int, [_P, _P, _I64, _I64, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P]),
    "mh_weight_norm_fwd": (ctypes.c_int, [_I32, _P, _P, _P, _P, _P, _P]),
    "mh_weight_norm_bwd": (ctypes.c_int, [_I32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mh_adam_step": (ctypes.c_int, [_P, _P, _P, _P, _I64, _I32, _P, _P, _P, _F, _F, _F, _P]),
    "mh_adam_step_dev": (ctypes.c_int, [_P, _P, _P, _P, _I64, _I32, _P, _P, _P, _P, _P, _F, _F, _F, _P]),
    "mh_masked_mean_workspace_floats": (_I64, []),
    "mh_masked_mean_fwd": (ctypes.c_int, [_I32, _P, _P, _P, _I64, _I32, _P, _P, _P, _P]),
    "mh_masked_mean_bwd": (ctypes.c_int, [_I32, _P, _P, _P, _I64, _I32, _P, _P, _P, _P, _P, _P]),
    "mh_ortho_perturb_fwd": (ctypes.c_int, [_P, _P, _P, _F, _I64, _P, _P]),
    "mh_ortho_perturb_bwd": (ctypes.c_int, [_P, _P, _P, _F, _I64, _P, _P]),
    "mh_smooth_points_fwd": (ctypes.c_int, [_P, _P, _P, _P, _I64, _I32, _P, _P, _P]),
    "mh_smooth_points_bwd": (ctypes.c_int, [_P, _P, _P, _P, _I64, _I32, _P, _P, _P, _P]),
    "mh_bg_blend_fwd": (ctypes.c_int, [_P, _P, _P, _I64, _P, _P]),
    "mh_bg_blend_bwd": (ctypes.c_int, [_P, _P, _P, _I64, _P, _P, _P]),
    "mh_pose_bwd_workspace_floats": (_I64, [_I64, _I64]),
    "mh_pose_apply_fwd": (ctypes.c_int, [_P, _P, _P, _P, _I64, _I64, _P, _P, _P]),
    "mh_pose_apply_bwd": (ctypes.c_int, [_P, _P, _P, _I64, _I64, _I64, _P, _P, _P, _P, _P]),
    "mh_render_loss_fwd": (ctypes.c_int, [_P] * 9 + [_I64, _F, _F, _F, _P, _P, _P, _P]),
    "mh_render_loss_bwd": (ctypes.c_int, [_P] * 7 + [_I64, _F, _F, _F, _P, _P, _P, _P, _P]),
    "mh_graph_count_memset_nodes": (ctypes.c_int, [_P, _P, _P, _P]),
    "mh_graph_replace_memset_nodes": (ctypes.c_int, [_P, _P]),
}

EXPORTS = tuple(_SIGS)
_lib = None


class MorpheusHipError(RuntimeError):
    pass


def load():
    """Load libmorpheus_hip.so; never falls back to anything else."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            raise MorpheusHipError(
                f"{SO} not found: build it with `python -m morpheus_amd.build` (hipcc, gfx950). "
                "The hot path has no CPU/PyTorch fallback by design.")
        lib = ctypes.CDLL(SO)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.mh_abi_version() != 8:
            raise MorpheusHipError("libmorpheus_hip.so ABI version mismatch")
        if os.environ.get("MORPHEUS_GRID_STAGE_MIN_POINTS"):       # tuning knob, see include/morpheus_hip.h
            lib.mh_grid_stage_min_points(int(os.environ["MORPHEUS_GRID_STAGE_MIN_POINTS"]))
        _lib = lib
    return _lib


def ptr(t):
    """Device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "C ABI takes contiguous buffers"
    return t.data_ptr()


# torch.cuda.current_stream() builds a Stream object through three Python layers (device-index resolution, an availability check
# that reads os.environ, Stream.__new__): ~7 us, once per C-ABI call, ~100 calls per eager real-view step whose host time IS the
# step time (tools/gpu/host_profile.py).  The raw handle of the same stream, when this torch has the accessor:
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_CUR_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    if _RAW_STREAM is not None and _CUR_DEVICE is not None:
        return _RAW_STREAM(_CUR_DEVICE())
    return torch.cuda.current_stream().cuda_stream


def check(status: int, what: str):
    if status != 0:
        raise MorpheusHipError(f"{what}: {load().mh_status_string(status).decode()} (status {status})")


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise MorpheusHipError("morpheus_amd ops run on an MI355X only (tensor is on %s); there is no CPU path"
                                   % t.device)
