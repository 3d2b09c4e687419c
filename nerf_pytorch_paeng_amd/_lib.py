"""ctypes binding of libmi_nerf.so (include/mi_nerf.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, a
``MiNerfError`` is raised.  The product path never routes through PyTorch ops or the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
# MI_NERF_LIB: an A/B variant built by `python -m nerf_pytorch_paeng_amd.build --variant TAG ...` (same ABI, same checks)
LIB_PATH = os.environ.get("MI_NERF_LIB") or os.path.join(HERE, "libmi_nerf.so")
ABI_VERSION = 4


class MiNerfError(RuntimeError):
    pass


class Net(C.Structure):          # mi_nerf_net
    _fields_ = [("D", C.c_int32), ("W", C.c_int32), ("skip", C.c_int32), ("L_x", C.c_int32), ("L_d", C.c_int32)]


class Params(C.Structure):       # mi_nerf_params
    _fields_ = [("linear_x_w", C.POINTER(C.c_void_p)), ("linear_x_b", C.POINTER(C.c_void_p)),
                ("linear_density_w", C.c_void_p), ("linear_density_b", C.c_void_p),
                ("linear_feat_w", C.c_void_p), ("linear_feat_b", C.c_void_p),
                ("linear_d_w", C.c_void_p), ("linear_d_b", C.c_void_p),
                ("linear_color_w", C.c_void_p), ("linear_color_b", C.c_void_p)]


class RenderCfg(C.Structure):    # mi_nerf_render_cfg
    _fields_ = [("near_", C.c_float), ("far_", C.c_float), ("Sc", C.c_int32), ("Nf", C.c_int32),
                ("det", C.c_int32), ("mode", C.c_int32), ("seed", C.c_uint32), ("reserved", C.c_uint32), ("ray_offset", C.c_int64)]


class WorkspaceLayout(C.Structure):   # mi_nerf_workspace_layout
    _fields_ = [("z_c", C.c_size_t), ("raw_c", C.c_size_t), ("weights_c", C.c_size_t), ("z_f", C.c_size_t),
                ("raw_f", C.c_size_t), ("total", C.c_size_t)]


class TrainLayout(C.Structure):       # mi_nerf_train_layout
    _fields_ = [(n, C.c_size_t) for n in ("stash_h", "stash_f", "stash_g", "mask_h", "mask_g", "stash_bytes", "delta_h", "delta_f",
                                          "delta_d", "emb", "partial", "work_bytes")]


_P, _I, _I64, _F, _U32, _SZ = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint32, C.c_size_t
_NETP = C.POINTER(Net)
_CFGP = C.POINTER(RenderCfg)

# name -> (restype, argtypes); mirrors include/mi_nerf.h declaration by declaration
SIGNATURES = {
    "mi_nerf_abi_version": (_I, []),
    "mi_nerf_last_error": (C.c_char_p, []),
    "mi_nerf_packed_bytes": (_SZ, [_NETP]),
    "mi_nerf_pack_weights": (_I, [_NETP, C.POINTER(Params), _P, _SZ]),
    "mi_nerf_packed_bytes_bf16": (_SZ, [_NETP]),
    "mi_nerf_pack_weights_bf16": (_I, [_NETP, C.POINTER(Params), _P, _SZ]),
    "mi_nerf_make_o_d": (_I, [_I, _I, C.POINTER(_F), C.POINTER(_F), _I, _I, _P, _P, _P]),
    "mi_nerf_make_o_d_pixels": (_I, [_I, _I, C.POINTER(_F), C.POINTER(_F), _P, _I64, _P, _P, _P]),
    "mi_nerf_ndc_rays": (_I, [_I, _I, _F, _F, _P, _I64, _P, _I64, _I64, _P, _P, _P]),
    "mi_nerf_fill_uniform": (_I, [_U32, _U32, _I64, _I64, _I, _P, _P]),
    "mi_nerf_stratified_z": (_I, [_I64, _I, _F, _F, _P, _P, _P]),
    "mi_nerf_sample_pdf": (_I, [_P, _P, _I64, _I, _I, _I, _P, _P, _P]),
    "mi_nerf_fine_z": (_I, [_P, _P, _I64, _I, _I, _I, _P, _P, _P, _P]),
    "mi_nerf_embed": (_I, [_P, _P, _I64, _I, _I, _I, _P, _P]),
    "mi_nerf_posenc": (_I, [_P, _I64, _I, _P, _P]),
    "mi_nerf_mlp_embedded": (_I, [_NETP, _P, _P, _I64, _P, _P]),
    "mi_nerf_mlp_rays": (_I, [_NETP, _P, _P, _P, _I64, _I, _P, _P]),
    "mi_nerf_mlp_rays_bf16": (_I, [_NETP, _P, _P, _P, _I64, _I, _P, _P]),
    "mi_nerf_mlp_rays_bf16_shape": (_I, [_NETP, _P, _P, _P, _I64, _I, _P, _I, _P]),
    "mi_nerf_packed_bytes_f16s": (_SZ, [_NETP]),
    "mi_nerf_pack_weights_f16s": (_I, [_NETP, C.POINTER(Params), _P, _SZ]),
    "mi_nerf_mlp_rays_f16s": (_I, [_NETP, _P, _P, _P, _I64, _I, _P, _P]),
    "mi_nerf_composite": (_I, [_P, _P, _P, _I, _I64, _I, _P, _P, _P, _P, _P, _P]),
    "mi_nerf_render_workspace_bytes": (_SZ, [_CFGP, _I64]),
    "mi_nerf_render_rays": (_I, [_NETP, _P, _P, _CFGP, _P, _I64, _P, _P, _P, _SZ, _P, _P, _P, _P, _P]),
    "mi_nerf_render_workspace_layout": (_I, [_CFGP, _I64, C.POINTER(WorkspaceLayout)]),
    "mi_nerf_composite_backward": (_I, [_P, _P, _P, _I, _I64, _I, _P, _P, _P]),
    "mi_nerf_param_count": (_SZ, [_NETP]),
    "mi_nerf_packed_bytes_bwd": (_SZ, [_NETP]),
    "mi_nerf_pack_weights_bwd": (_I, [_NETP, C.POINTER(Params), _P, _SZ]),
    "mi_nerf_pack_map": (_I, [_NETP, _I, _P, _SZ]),
    "mi_nerf_pack_apply": (_I, [_P, _P, _SZ, _P, _P]),
    "mi_nerf_pack_map_bf16_len": (_SZ, [_NETP]),
    "mi_nerf_pack_map_bf16": (_I, [_NETP, _P, _SZ]),
    "mi_nerf_pack_apply_bf16": (_I, [_NETP, _P, _P, _P, _SZ, _P]),
    "mi_nerf_train_layout_query": (_I, [_NETP, _I64, _I, C.POINTER(TrainLayout)]),
    "mi_nerf_mlp_rays_train": (_I, [_NETP, _P, _P, _P, _I64, _I, _P, _P, _SZ, _P]),
    "mi_nerf_mlp_rays_train_f16s": (_I, [_NETP, _P, _P, _P, _I64, _I, _P, _P, _SZ, _P]),
    "mi_nerf_pack_map_f16s_len": (_SZ, [_NETP]),
    "mi_nerf_pack_map_f16s": (_I, [_NETP, _P, _SZ]),
    "mi_nerf_pack_apply_f16s": (_I, [_NETP, _P, _P, _P, _SZ, _P, _P]),
    "mi_nerf_packed_bytes_bwd_f16s": (_SZ, [_NETP]),
    "mi_nerf_pack_weights_bwd_f16s": (_I, [_NETP, C.POINTER(Params), _P, _SZ]),
    "mi_nerf_pack_map_bwd_f16s_len": (_SZ, [_NETP]),
    "mi_nerf_pack_map_bwd_f16s": (_I, [_NETP, _P, _SZ]),
    "mi_nerf_pack_apply_bwd_f16s": (_I, [_NETP, _P, _P, _P, _SZ, _P, _P]),
    "mi_nerf_mlp_backward": (_I, [_NETP, _P, _P, _P, _P, _I64, _I, _P, _P, _P, _SZ, _P, _I, _P]),
    "mi_nerf_mlp_backward_mode": (_I, [_NETP, _P, _P, _P, _P, _I64, _I, _P, _P, _P, _SZ, _P, _I, _I, _P]),
    "mi_nerf_mlp_embedded_train": (_I, [_NETP, _P, _P, _I64, _P, _P, _SZ, _P]),
    "mi_nerf_mlp_embedded_backward": (_I, [_NETP, _P, _P, _P, _I64, _P, _P, _P, _SZ, _P, _P]),
    "mi_nerf_wgrad_scratch_bytes": (_SZ, []),
    "mi_nerf_wgrad_products": (_I, [_I, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _P, _SZ, _I, C.POINTER(_F), _P]),
    "mi_nerf_wgrad_products_f16s": (_I, [_I, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _P, _SZ, _I, C.POINTER(_F), _P]),
    "mi_nerf_image_metrics": (_I, [_P, _P, _I64, _P, _P, _SZ, _P]),
    "mi_nerf_nanmax": (_I, [_P, _I64, _P, _P, _SZ, _P]),
    "mi_nerf_to8b": (_I, [_P, _I64, _P, _P, _P]),
    "mi_nerf_rays_rgb": (_I, [_I, _I, C.POINTER(_F), _P, _P, _I64, _P, _P]),
    "mi_nerf_permute_rows": (_I, [_P, _P, _I64, _I, _P, _P]),
    "mi_nerf_rccl_available": (_I, []),
    "mi_nerf_comm_unique_id": (_I, [_P]),
    "mi_nerf_comm_init_rank": (_I, [_P, _I, _I, C.POINTER(_P)]),
    "mi_nerf_comm_info": (_I, [_P, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "mi_nerf_comm_destroy": (_I, [_P]),
    "mi_nerf_all_gather_staging_bytes": (_SZ, [_I, _I, _I, _I]),
    "mi_nerf_all_gather_tiles": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _SZ, _P]),
    "mi_nerf_unpad_tiles": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "mi_nerf_time_mlp_rays": (_I, [_NETP, _P, _P, _P, _I64, _I, _P, _I, _I, C.POINTER(_F), _P]),
    "mi_nerf_selftest_mfma": (_I, [_P]),
}

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raise loudly if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MiNerfError(
                f"{LIB_PATH} not found: build it with `python -m nerf_pytorch_paeng_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU/PyTorch fallback for this path.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise MiNerfError(f"{LIB_PATH} does not export {name}: stale build?") from e
            fn.restype, fn.argtypes = res, args
        v = handle.mi_nerf_abi_version()
        if v != ABI_VERSION:
            raise MiNerfError(f"ABI mismatch: library {v}, binding {ABI_VERSION}")
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().mi_nerf_last_error()
        raise MiNerfError(f"{what} failed (status {rc}): {msg.decode() if msg else '?'}")


def dev_ptr(t: Optional[torch.Tensor], name: str = "tensor", dtype=torch.float32, align: int = 4) -> Optional[int]:
    """data_ptr() of a contiguous device tensor of the expected dtype (None passes through as NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise MiNerfError(f"{name} must live on a HIP device (got {t.device}); this path has no CPU fallback")
    if t.dtype != dtype:
        raise MiNerfError(f"{name} must be {dtype} (got {t.dtype})")
    if not t.is_contiguous():
        raise MiNerfError(f"{name} must be contiguous")
    p = t.data_ptr()
    if p % align:
        raise MiNerfError(f"{name} must be {align}-byte aligned")
    return p


def stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def as_f32_dev(t, device=None) -> torch.Tensor:
    """Contiguous fp32 tensor on `device` (no copy when already so)."""
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(t)
    if device is not None and t.device != torch.device(device):
        t = t.to(device)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()
