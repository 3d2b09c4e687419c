"""Tensor-level wrappers over the C ABI: allocate outputs with torch, hand raw pointers across.

Every function launches on ``torch.cuda.current_stream`` of the tensors' device and returns
immediately (no host synchronisation).  PyTorch is used for device memory and streams only.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from ._lib import MiNerfError, Net, RenderCfg, TrainLayout, WorkspaceLayout, check, dev_ptr, lib, stream_ptr


def _guard(device):
    device = torch.device(device)
    if device.type != "cuda":
        raise MiNerfError(f"tensors must live on a HIP device (got {device}); this path has no CPU fallback")
    return torch.cuda.device(device)


def make_net(D: int, W: int, skip: int = 4, L_x: int = 10, L_d: int = 4) -> Net:
    return Net(D, W, skip, L_x, L_d)


# ------------------------------------------------------------------------------------------------
# weights
# ------------------------------------------------------------------------------------------------
def pack_module(sd: Dict[str, "np.ndarray | torch.Tensor"], prefix: str, net: Net, bf16: bool = False, backward: bool = False,
                f16s: bool = False) -> torch.Tensor:
    """Pack one NeRFModule (keys ``{prefix}linear_x.{i}.weight`` ...; model/NeRF.py:24-30) into the
    kernels' streaming layout (``backward``: the transposed stream of the backward-data kernel).
    Returns a CPU uint8 tensor; copy it to the device once."""
    def arr(key):
        v = sd[prefix + key]
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        return np.ascontiguousarray(v, dtype=np.float32)

    in_x, in_d, W, D = 3 + 6 * net.L_x, 3 + 6 * net.L_d, net.W, net.D
    keep = []
    def ptr(a, shape):
        if tuple(a.shape) != tuple(shape):
            raise MiNerfError(f"parameter shape {a.shape} != expected {shape}")
        keep.append(a)
        return a.ctypes.data_as(C.c_void_p)

    wx = (C.c_void_p * D)()
    bx = (C.c_void_p * D)()
    for l in range(D):
        fan_in = in_x if l == 0 else (W + in_x if (net.skip >= 0 and l == net.skip + 1) else W)
        wx[l] = ptr(arr(f"linear_x.{l}.weight"), (W, fan_in))
        bx[l] = ptr(arr(f"linear_x.{l}.bias"), (W,))
    p = L.Params(wx, bx,
                 ptr(arr("linear_density.weight"), (1, W)), ptr(arr("linear_density.bias"), (1,)),
                 ptr(arr("linear_feat.weight"), (W, W)), ptr(arr("linear_feat.bias"), (W,)),
                 ptr(arr("linear_d.weight"), (W // 2, W + in_d)), ptr(arr("linear_d.bias"), (W // 2,)),
                 ptr(arr("linear_color.weight"), (3, W // 2)), ptr(arr("linear_color.bias"), (3,)))
    if backward and f16s:
        size_fn, pack_fn = lib().mi_nerf_packed_bytes_bwd_f16s, lib().mi_nerf_pack_weights_bwd_f16s
    elif backward:
        size_fn, pack_fn = lib().mi_nerf_packed_bytes_bwd, lib().mi_nerf_pack_weights_bwd
    elif f16s:
        size_fn, pack_fn = lib().mi_nerf_packed_bytes_f16s, lib().mi_nerf_pack_weights_f16s
    elif bf16:
        size_fn, pack_fn = lib().mi_nerf_packed_bytes_bf16, lib().mi_nerf_pack_weights_bf16
    else:
        size_fn, pack_fn = lib().mi_nerf_packed_bytes, lib().mi_nerf_pack_weights
    nbytes = size_fn(C.byref(net))
    if nbytes == 0:
        check(1, "mi_nerf_packed_bytes")
    blob = torch.empty(nbytes, dtype=torch.uint8)
    check(pack_fn(C.byref(net), C.byref(p), blob.data_ptr(), nbytes), "mi_nerf_pack_weights")
    return blob


# ------------------------------------------------------------------------------------------------
# rays
# ------------------------------------------------------------------------------------------------
def _cam(K, pose) -> Tuple[C.Array, C.Array]:
    k = K.detach().cpu().numpy() if isinstance(K, torch.Tensor) else np.asarray(K)
    # torch rounds the float64 intrinsics to fp32 where they meet the fp32 pixel grid (rays.py:28-29)
    k4 = (C.c_float * 4)(float(np.float32(k[0][0])), float(np.float32(k[1][1])), float(np.float32(k[0][2])), float(np.float32(k[1][2])))
    p = pose.detach().cpu().numpy() if isinstance(pose, torch.Tensor) else np.asarray(pose)
    p = np.asarray(p, dtype=np.float32)[:3, :4]
    return k4, (C.c_float * 12)(*p.reshape(-1).tolist())


def make_o_d(W: int, H: int, K, pose, device, row0: int = 0, n_rows: Optional[int] = None, want_origins: bool = True):
    n_rows = H - row0 if n_rows is None else n_rows
    device = torch.device(device)
    k4, p12 = _cam(K, pose)
    d = torch.empty(n_rows, W, 3, dtype=torch.float32, device=device)
    o = torch.empty(n_rows, W, 3, dtype=torch.float32, device=device) if want_origins else None
    with _guard(device):
        check(lib().mi_nerf_make_o_d(W, H, k4, p12, row0, n_rows, dev_ptr(o), dev_ptr(d), stream_ptr(device)), "mi_nerf_make_o_d")
    return o, d


def make_o_d_pixels(W: int, H: int, K, pose, pix: torch.Tensor):
    device = pix.device
    k4, p12 = _cam(K, pose)
    n = pix.numel()
    o = torch.empty(n, 3, dtype=torch.float32, device=device)
    d = torch.empty(n, 3, dtype=torch.float32, device=device)
    with _guard(device):
        check(lib().mi_nerf_make_o_d_pixels(W, H, k4, p12, dev_ptr(pix, "pix", torch.int64, 8), n, dev_ptr(o), dev_ptr(d),
                                            stream_ptr(device)), "mi_nerf_make_o_d_pixels")
    return o, d


def ndc_rays(H: int, W: int, focal: float, near: float, rays_o: torch.Tensor, rays_d: torch.Tensor):
    device = rays_d.device
    n = rays_d.shape[0]
    # accept the stride-0 expanded origin view make_o_d returns (rays.py:33) without materialising it
    def strided(t, name):
        if t.dim() != 2 or t.shape[1] != 3 or t.dtype != torch.float32 or not t.is_cuda:
            raise MiNerfError(f"{name} must be a [n,3] fp32 device tensor")
        if t.stride(1) != 1 or (t.stride(0) not in (0, 3) and n > 1):
            t = t.contiguous()
        return t, (t.stride(0) if n > 1 else 3)
    o, os_ = strided(rays_o, "rays_o")
    d, ds_ = strided(rays_d, "rays_d")
    oo = torch.empty(n, 3, dtype=torch.float32, device=device)
    dd = torch.empty(n, 3, dtype=torch.float32, device=device)
    with _guard(device):
        check(lib().mi_nerf_ndc_rays(H, W, float(focal), float(near), o.data_ptr(), os_, d.data_ptr(), ds_, n, dev_ptr(oo),
                                     dev_ptr(dd), stream_ptr(device)), "mi_nerf_ndc_rays")
    return oo, dd


# ------------------------------------------------------------------------------------------------
# sampling
# ------------------------------------------------------------------------------------------------
def fill_uniform(seed: int, stream_id: int, ray0: int, n_rays: int, n_samples: int, device, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    device = torch.device(device)
    if out is None:
        out = torch.empty(n_rays, n_samples, dtype=torch.float32, device=device)
    else:
        if tuple(out.shape) != (n_rays, n_samples):
            raise MiNerfError(f"out must be [{n_rays}, {n_samples}], got {tuple(out.shape)}")
        device = out.device
    with _guard(device):
        check(lib().mi_nerf_fill_uniform(seed & 0xFFFFFFFF, stream_id, ray0, n_rays, n_samples, dev_ptr(out), stream_ptr(device)),
              "mi_nerf_fill_uniform")
    return out


def stratified_z(near: float, far: float, t_rand: torch.Tensor) -> torch.Tensor:
    n, S = t_rand.shape
    z = torch.empty_like(t_rand)
    with _guard(t_rand.device):
        check(lib().mi_nerf_stratified_z(n, S, float(near), float(far), dev_ptr(t_rand, "t_rand"), dev_ptr(z),
                                         stream_ptr(t_rand.device)), "mi_nerf_stratified_z")
    return z


def sample_pdf(bins: torch.Tensor, weights: torch.Tensor, N: int, det: bool, u: Optional[torch.Tensor]) -> torch.Tensor:
    n, B = bins.shape
    if tuple(weights.shape) != (n, B - 1):
        raise MiNerfError(f"weights must be [n, B-1] = {(n, B - 1)}, got {tuple(weights.shape)}")
    if not det and (u is None or tuple(u.shape) != (n, N)):
        raise MiNerfError("u [n, N] is required unless det")
    out = torch.empty(n, N, dtype=torch.float32, device=bins.device)
    with _guard(bins.device):
        check(lib().mi_nerf_sample_pdf(dev_ptr(bins, "bins"), dev_ptr(weights, "weights"), n, B, N, int(det),
                                       None if det else dev_ptr(u, "u"), dev_ptr(out), stream_ptr(bins.device)), "mi_nerf_sample_pdf")
    return out


def fine_z(z_c: torch.Tensor, weights_c: torch.Tensor, Nf: int, det: bool, u: Optional[torch.Tensor], want_samples: bool = False):
    n, Sc = z_c.shape
    if tuple(weights_c.shape) != (n, Sc):
        raise MiNerfError("weights_c must match z_c")
    if not det and (u is None or tuple(u.shape) != (n, Nf)):
        raise MiNerfError("u [n, Nf] is required unless det")
    z_f = torch.empty(n, Sc + Nf, dtype=torch.float32, device=z_c.device)
    zs = torch.empty(n, Nf, dtype=torch.float32, device=z_c.device) if want_samples else None
    with _guard(z_c.device):
        check(lib().mi_nerf_fine_z(dev_ptr(z_c, "z_c"), dev_ptr(weights_c, "weights_c"), n, Sc, Nf, int(det),
                                   None if det else dev_ptr(u, "u"), dev_ptr(z_f), dev_ptr(zs), stream_ptr(z_c.device)), "mi_nerf_fine_z")
    return (z_f, zs) if want_samples else z_f


# ------------------------------------------------------------------------------------------------
# encoding, network, compositing
# ------------------------------------------------------------------------------------------------
def embed(rays: torch.Tensor, z: torch.Tensor, L_x: int, L_d: int) -> torch.Tensor:
    n, S = z.shape
    out = torch.empty(n * S, 6 + 6 * L_x + 6 * L_d, dtype=torch.float32, device=z.device)
    with _guard(z.device):
        check(lib().mi_nerf_embed(dev_ptr(rays, "rays"), dev_ptr(z, "z"), n, S, L_x, L_d, dev_ptr(out), stream_ptr(z.device)), "mi_nerf_embed")
    return out


def posenc(x: torch.Tensor, Lf: int) -> torch.Tensor:
    n = x.shape[0]
    out = torch.empty(n, 3 + 6 * Lf, dtype=torch.float32, device=x.device)
    with _guard(x.device):
        check(lib().mi_nerf_posenc(dev_ptr(x, "x"), n, Lf, dev_ptr(out), stream_ptr(x.device)), "mi_nerf_posenc")
    return out


def mlp_embedded(net: Net, packed: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    n = x.shape[0]
    if x.dim() != 2 or x.shape[1] != 6 + 6 * net.L_x + 6 * net.L_d:
        raise MiNerfError(f"x must be [n, {6 + 6 * net.L_x + 6 * net.L_d}], got {tuple(x.shape)}")
    out = torch.empty(n, 4, dtype=torch.float32, device=x.device)
    with _guard(x.device):
        check(lib().mi_nerf_mlp_embedded(C.byref(net), dev_ptr(packed, "packed", torch.uint8, 16), dev_ptr(x, "x"), n, dev_ptr(out, "out", align=16),
                                         stream_ptr(x.device)), "mi_nerf_mlp_embedded")
    return out


def mlp_rays(net: Net, packed: torch.Tensor, rays: torch.Tensor, z: torch.Tensor, bf16: bool = False, points_per_wave: int = 0,
             f16s: bool = False) -> torch.Tensor:
    """Fused encoding + MLP over rays x depths.  ``points_per_wave`` (bf16 only): 0 = launch shape chosen per launch, 64 / 32 pinned.
    ``f16s``: the split-precision variant (fp32-grade results on the f16 matrix pipe; ``packed`` from pack_module(..., f16s=True))."""
    n, S = z.shape
    if tuple(rays.shape) != (n, 6):
        raise MiNerfError(f"rays must be [n,6], got {tuple(rays.shape)}")
    raw = torch.empty(n, S, 4, dtype=torch.float32, device=z.device)
    args = (C.byref(net), dev_ptr(packed, "packed", torch.uint8, 16), dev_ptr(rays, "rays"), dev_ptr(z, "z"), n, S, dev_ptr(raw, "raw", align=16))
    with _guard(z.device):
        if f16s:
            check(lib().mi_nerf_mlp_rays_f16s(*args, stream_ptr(z.device)), "mi_nerf_mlp_rays_f16s")
        elif bf16:
            check(lib().mi_nerf_mlp_rays_bf16_shape(*args, int(points_per_wave), stream_ptr(z.device)), "mi_nerf_mlp_rays_bf16_shape")
        else:
            check(lib().mi_nerf_mlp_rays(*args, stream_ptr(z.device)), "mi_nerf_mlp_rays")
    return raw


def composite(raw: torch.Tensor, z: torch.Tensor, rays_or_d: torch.Tensor, want_all: bool = True):
    n, S = z.shape
    if tuple(raw.shape) != (n, S, 4):
        raise MiNerfError(f"raw must be [n,S,4] = {(n, S, 4)}, got {tuple(raw.shape)}")
    stride = rays_or_d.shape[-1]
    if tuple(rays_or_d.shape) != (n, stride) or stride not in (3, 6):
        raise MiNerfError("rays must be [n,6] or rays_d [n,3]")
    dev = z.device
    rgb = torch.empty(n, 3, dtype=torch.float32, device=dev)
    disp = torch.empty(n, dtype=torch.float32, device=dev)
    acc = torch.empty(n, dtype=torch.float32, device=dev) if want_all else None
    wts = torch.empty(n, S, dtype=torch.float32, device=dev) if want_all else None
    depth = torch.empty(n, dtype=torch.float32, device=dev) if want_all else None
    with _guard(dev):
        check(lib().mi_nerf_composite(dev_ptr(raw, "raw", align=16), dev_ptr(z, "z"), dev_ptr(rays_or_d, "rays"), stride, n, S, dev_ptr(rgb),
                                      dev_ptr(disp), dev_ptr(acc), dev_ptr(wts), dev_ptr(depth), stream_ptr(dev)), "mi_nerf_composite")
    return rgb, disp, acc, wts, depth


def composite_backward(raw: torch.Tensor, z: torch.Tensor, rays_or_d: torch.Tensor, d_rgb: torch.Tensor) -> torch.Tensor:
    """d rgb_map [n,3] -> d raw [n,S,4]: what autograd yields for nerf_process.py:89-140 when the loss reads rgb_map."""
    n, S = z.shape
    if tuple(raw.shape) != (n, S, 4) or tuple(d_rgb.shape) != (n, 3):
        raise MiNerfError(f"raw must be {(n, S, 4)} and d_rgb {(n, 3)}, got {tuple(raw.shape)} / {tuple(d_rgb.shape)}")
    stride = rays_or_d.shape[-1]
    if tuple(rays_or_d.shape) != (n, stride) or stride not in (3, 6):
        raise MiNerfError("rays must be [n,6] or rays_d [n,3]")
    d_raw = torch.empty(n, S, 4, dtype=torch.float32, device=z.device)
    with _guard(z.device):
        check(lib().mi_nerf_composite_backward(dev_ptr(raw, "raw", align=16), dev_ptr(z, "z"), dev_ptr(rays_or_d, "rays"), stride, n, S,
                                               dev_ptr(d_rgb, "d_rgb"), dev_ptr(d_raw, "d_raw", align=16), stream_ptr(z.device)),
              "mi_nerf_composite_backward")
    return d_raw


# ------------------------------------------------------------------------------------------------
# training path: stash forward, backward, device-side re-pack
# ------------------------------------------------------------------------------------------------
PARAM_ORDER = ("linear_x", "linear_d", "linear_feat", "linear_density", "linear_color")   # module.parameters() order, NeRF.py:24-30


def param_count(net: Net) -> int:
    n = lib().mi_nerf_param_count(C.byref(net))
    if n == 0:
        check(1, "mi_nerf_param_count")
    return int(n)


def param_names(net: Net):
    """state_dict keys of one NeRFModule in flat-vector order."""
    names = []
    for l in range(net.D):
        names += [f"linear_x.{l}.weight", f"linear_x.{l}.bias"]
    for head in PARAM_ORDER[1:]:
        names += [f"{head}.weight", f"{head}.bias"]
    return names


def flatten_params(sd, prefix: str, net: Net, device=None) -> torch.Tensor:
    """cat of the module's parameters in flat-vector order (fp32)."""
    flat = torch.cat([torch.as_tensor(sd[prefix + k]).detach().reshape(-1).float() for k in param_names(net)])
    if flat.numel() != param_count(net):
        raise MiNerfError(f"flat parameter vector has {flat.numel()} entries, expected {param_count(net)}")
    return flat.to(device) if device is not None else flat


def train_layout(net: Net, n_rays: int, S: int) -> TrainLayout:
    lay = TrainLayout()
    check(lib().mi_nerf_train_layout_query(C.byref(net), int(n_rays), int(S), C.byref(lay)), "mi_nerf_train_layout_query")
    return lay


def pack_map(net: Net, backward: bool = False) -> torch.Tensor:
    """Gather map (CPU int32) from the flat parameter vector to the forward / backward-data blob."""
    nbytes = (lib().mi_nerf_packed_bytes_bwd if backward else lib().mi_nerf_packed_bytes)(C.byref(net))
    if nbytes == 0:
        check(1, "mi_nerf_packed_bytes")
    m = torch.empty(nbytes // 4, dtype=torch.int32)
    check(lib().mi_nerf_pack_map(C.byref(net), int(backward), m.data_ptr(), m.numel()), "mi_nerf_pack_map")
    return m


def pack_apply(map_dev: torch.Tensor, flat: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Device-side pack: blob (uint8, 4*len(map) bytes) from the flat parameter vector."""
    dev = flat.device
    nbytes = map_dev.numel() * 4
    if out is None:
        out = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    if out.numel() != nbytes:
        raise MiNerfError(f"blob must have {nbytes} bytes, got {out.numel()}")
    with _guard(dev):
        check(lib().mi_nerf_pack_apply(dev_ptr(map_dev, "map", torch.int32), dev_ptr(flat, "flat"), nbytes, dev_ptr(out, "blob", torch.uint8, 16),
                                       stream_ptr(dev)), "mi_nerf_pack_apply")
    return out


def pack_map_bf16(net: Net) -> torch.Tensor:
    """Gather map (CPU int32) from the flat parameter vector to the bf16 blob: stream elements, then side-table floats."""
    n = lib().mi_nerf_pack_map_bf16_len(C.byref(net))
    if n == 0:
        check(1, "mi_nerf_pack_map_bf16_len")
    m = torch.empty(n, dtype=torch.int32)
    check(lib().mi_nerf_pack_map_bf16(C.byref(net), m.data_ptr(), m.numel()), "mi_nerf_pack_map_bf16")
    return m


def pack_apply_bf16(net: Net, map_dev: torch.Tensor, flat: torch.Tensor) -> torch.Tensor:
    """Device-side pack of the bf16 blob (uint8) from the flat parameter vector."""
    dev = flat.device
    nbytes = lib().mi_nerf_packed_bytes_bf16(C.byref(net))
    out = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with _guard(dev):
        check(lib().mi_nerf_pack_apply_bf16(C.byref(net), dev_ptr(map_dev, "map", torch.int32), dev_ptr(flat, "flat"), dev_ptr(out, "blob", torch.uint8, 16),
                                            nbytes, stream_ptr(dev)), "mi_nerf_pack_apply_bf16")
    return out


def pack_map_f16s(net: Net, backward: bool = False) -> torch.Tensor:
    """Gather map (CPU int32) from the flat parameter vector to the split-precision blob: stream elements (the source weight at its hi and
    at its lo position), then side-table floats (``backward``: the transposed stream, no side tables)."""
    n = (lib().mi_nerf_pack_map_bwd_f16s_len if backward else lib().mi_nerf_pack_map_f16s_len)(C.byref(net))
    if n == 0:
        check(1, "mi_nerf_pack_map_f16s_len")
    m = torch.empty(n, dtype=torch.int32)
    check((lib().mi_nerf_pack_map_bwd_f16s if backward else lib().mi_nerf_pack_map_f16s)(C.byref(net), m.data_ptr(), m.numel()), "mi_nerf_pack_map_f16s")
    return m


def pack_apply_f16s(net: Net, map_dev: torch.Tensor, flat: torch.Tensor, out_of_range: Optional[torch.Tensor] = None, backward: bool = False) -> torch.Tensor:
    """Device-side pack of the split-precision blob (uint8) from the flat parameter vector (``backward``: the transposed stream of the
    split-precision backward-data kernel, map from pack_map_f16s(net, backward=True)).  ``out_of_range`` (int32 [1] on the device, or None)
    counts stream elements whose weight is NaN or beyond the f16 range -- the host packer refuses those."""
    dev = flat.device
    nbytes = (lib().mi_nerf_packed_bytes_bwd_f16s if backward else lib().mi_nerf_packed_bytes_f16s)(C.byref(net))
    if nbytes == 0:
        check(1, "mi_nerf_packed_bytes_f16s")
    out = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    fn = lib().mi_nerf_pack_apply_bwd_f16s if backward else lib().mi_nerf_pack_apply_f16s
    with _guard(dev):
        check(fn(C.byref(net), dev_ptr(map_dev, "map", torch.int32), dev_ptr(flat, "flat"), dev_ptr(out, "blob", torch.uint8, 16),
                 nbytes, None if out_of_range is None else dev_ptr(out_of_range, "out_of_range", torch.int32), stream_ptr(dev)),
              "mi_nerf_pack_apply_f16s")
    return out


def mlp_rays_train(net: Net, packed: torch.Tensor, rays: torch.Tensor, z: torch.Tensor, stash: Optional[torch.Tensor] = None, f16s: bool = False):
    """Training forward: raw [n,S,4] plus the activation stash the backward reads.  ``f16s``: the forward runs in split precision (``packed``
    from pack_apply_f16s / pack_module(..., f16s=True)); the stash has the same tensors in the same layouts either way."""
    n, S = z.shape
    if tuple(rays.shape) != (n, 6):
        raise MiNerfError(f"rays must be [n,6], got {tuple(rays.shape)}")
    lay = train_layout(net, n, S)
    if stash is None:
        stash = torch.empty(lay.stash_bytes, dtype=torch.uint8, device=z.device)
    raw = torch.empty(n, S, 4, dtype=torch.float32, device=z.device)
    with _guard(z.device):
        fn = lib().mi_nerf_mlp_rays_train_f16s if f16s else lib().mi_nerf_mlp_rays_train
        check(fn(C.byref(net), dev_ptr(packed, "packed", torch.uint8, 16), dev_ptr(rays, "rays"), dev_ptr(z, "z"), n, S,
                 dev_ptr(raw, "raw", align=16), dev_ptr(stash, "stash", torch.uint8, 16), stash.numel(),
                 stream_ptr(z.device)), "mi_nerf_mlp_rays_train_f16s" if f16s else "mi_nerf_mlp_rays_train")
    return raw, stash


def mlp_backward(net: Net, packed: torch.Tensor, packed_bwd: torch.Tensor, rays: torch.Tensor, z: torch.Tensor, d_raw: torch.Tensor,
                 stash: torch.Tensor, work: Optional[torch.Tensor] = None, stage: int = 0, grads: Optional[torch.Tensor] = None,
                 f16s_wgrad: bool = False, f16s_dgrad: bool = False):
    """d_raw [n,S,4] -> flat parameter gradient (param_names order).  Returns (grads, work).
    ``f16s_wgrad``: the W-wide weight-gradient products run in split precision (fp32-grade results, HBM-bound instead of MFMA-bound).
    ``f16s_dgrad``: the backward-data chain runs in split precision; ``packed_bwd`` is then the blob of pack_apply_f16s(..., backward=True).
    The weight-gradient kernels write EVERY element of the flat vector (each parameter block is the output of exactly one product's
    reduction), so it is allocated uninitialised; ``grads`` lets a test pass a poisoned buffer to check exactly that.  Two calls write
    nothing of it and say so: ``stage=1`` (deltas only: ``grads`` comes back as None) and an empty batch (zero-filled by the library)."""
    n, S = z.shape
    if tuple(d_raw.shape) != (n, S, 4) or tuple(rays.shape) != (n, 6):
        raise MiNerfError(f"d_raw must be {(n, S, 4)} and rays {(n, 6)}, got {tuple(d_raw.shape)} / {tuple(rays.shape)}")
    lay = train_layout(net, n, S)
    dev = z.device
    if work is None:
        work = torch.empty(lay.work_bytes, dtype=torch.uint8, device=dev)
    if grads is None:
        grads = torch.empty(param_count(net), dtype=torch.float32, device=dev)
    elif grads.numel() != param_count(net):
        raise MiNerfError(f"grads must have {param_count(net)} elements, got {grads.numel()}")
    with _guard(dev):
        check(lib().mi_nerf_mlp_backward_mode(C.byref(net), dev_ptr(packed, "packed", torch.uint8, 16), dev_ptr(packed_bwd, "packed_bwd", torch.uint8, 16),
                                              dev_ptr(rays, "rays"), dev_ptr(z, "z"), n, S, dev_ptr(d_raw, "d_raw", align=16),
                                              dev_ptr(stash, "stash", torch.uint8, 16), dev_ptr(work, "work", torch.uint8, 16), work.numel(),
                                              dev_ptr(grads, "grads"), int(stage), (1 if f16s_wgrad else 0) | (2 if f16s_dgrad else 0), stream_ptr(dev)), "mi_nerf_mlp_backward_mode")
    return (None if stage == 1 else grads), work


def mlp_embedded_train(net: Net, packed: torch.Tensor, x: torch.Tensor):
    """Training forward over pre-embedded rows [n, in_x + in_d]: (out [n,4], stash)."""
    n = x.shape[0]
    if x.dim() != 2 or x.shape[1] != 6 + 6 * net.L_x + 6 * net.L_d:
        raise MiNerfError(f"x must be [n, {6 + 6 * net.L_x + 6 * net.L_d}], got {tuple(x.shape)}")
    lay = train_layout(net, (n + 31) // 32, 32)
    stash = torch.empty(lay.stash_bytes, dtype=torch.uint8, device=x.device)
    out = torch.empty(n, 4, dtype=torch.float32, device=x.device)
    with _guard(x.device):
        check(lib().mi_nerf_mlp_embedded_train(C.byref(net), dev_ptr(packed, "packed", torch.uint8, 16), dev_ptr(x, "x"), n, dev_ptr(out, "out", align=16),
                                               dev_ptr(stash, "stash", torch.uint8, 16), stash.numel(), stream_ptr(x.device)), "mi_nerf_mlp_embedded_train")
    return out, stash


def mlp_embedded_backward(net: Net, packed: torch.Tensor, packed_bwd: torch.Tensor, x: torch.Tensor, d_out: torch.Tensor, stash: torch.Tensor,
                          grads: Optional[torch.Tensor] = None) -> torch.Tensor:
    """d_out [n,4] -> flat parameter gradient (param_names order) for the embedded-row forward (every element written, see mlp_backward)."""
    n = x.shape[0]
    if tuple(d_out.shape) != (n, 4):
        raise MiNerfError(f"d_out must be {(n, 4)}, got {tuple(d_out.shape)}")
    lay = train_layout(net, (n + 31) // 32, 32)
    dev = x.device
    work = torch.empty(lay.work_bytes, dtype=torch.uint8, device=dev)
    if grads is None:
        grads = torch.empty(param_count(net), dtype=torch.float32, device=dev)
    with _guard(dev):
        check(lib().mi_nerf_mlp_embedded_backward(C.byref(net), dev_ptr(packed, "packed", torch.uint8, 16), dev_ptr(packed_bwd, "packed_bwd", torch.uint8, 16),
                                                  dev_ptr(x, "x"), n, dev_ptr(d_out, "d_out", align=16), dev_ptr(stash, "stash", torch.uint8, 16),
                                                  dev_ptr(work, "work", torch.uint8, 16), work.numel(), dev_ptr(grads, "grads"), stream_ptr(dev)),
              "mi_nerf_mlp_embedded_backward")
    return grads


def wgrad_product(delta: torch.Tensor, M: int, x: torch.Tensor, N: int, P: int, want_bias: bool = True, iters: int = 1, timed: bool = False):
    """out[M,N] = delta[:P,:M]^T x[:P,:N] (+ bias[M] = column sums of delta): ``wgrad_products`` with one product (the C ABI has no
    single-product entry since round 4).  Returns (out, bias, avg_ms) -- avg_ms only when ``timed``."""
    outs, biases, ms = wgrad_products([delta], [x], P, [int(M)], [int(N)], want_bias, iters, timed)
    return outs[0], (biases[0] if want_bias else None), ms


def wgrad_products(deltas: Sequence[torch.Tensor], xs: Sequence[torch.Tensor], P: int, Ms: Optional[Sequence[int]] = None,
                   Ns: Optional[Sequence[int]] = None, want_bias: bool = True, iters: int = 1, timed: bool = False, f16s: bool = False):
    """Several products over the same P points (mi_nerf_wgrad_products): out[b] = deltas[b][:P, :M_b]^T xs[b][:P, :N_b]
    (+ bias[b] = column sums of deltas[b]).  The wide ones (both sides wider than 64 columns) share ONE launch -- how the backward pass
    runs a network's 256 x 256 products -- a product with a narrow side runs in a launch of its own behind them.  ``f16s``: in split
    precision (mi_nerf_wgrad_products_f16s; fp32-grade results, bound by the operands' HBM reads).  Returns (outs, biases, avg_ms)."""
    n = len(deltas)
    if n == 0 or len(xs) != n:
        raise MiNerfError("deltas / xs must be non-empty lists of equal length")
    dev = deltas[0].device
    Ms = [d.shape[1] for d in deltas] if Ms is None else list(Ms)
    Ns = [x.shape[1] for x in xs] if Ns is None else list(Ns)
    for d, x, M, N in zip(deltas, xs, Ms, Ns):
        if d.dim() != 2 or x.dim() != 2 or d.shape[0] < P or x.shape[0] < P or d.shape[1] < M or x.shape[1] < N:
            raise MiNerfError(f"operands {tuple(d.shape)} / {tuple(x.shape)} too small for P={P}, M={M}, N={N}")
    outs = [torch.empty(M, N, dtype=torch.float32, device=dev) for M, N in zip(Ms, Ns)]
    biases = [torch.empty(M, dtype=torch.float32, device=dev) for M in Ms] if want_bias else None
    scratch = torch.empty(int(lib().mi_nerf_wgrad_scratch_bytes()), dtype=torch.uint8, device=dev)
    PP, II = C.c_void_p * n, C.c_int * n
    ms = C.c_float(0.0)
    with _guard(dev):
        check((lib().mi_nerf_wgrad_products_f16s if f16s else lib().mi_nerf_wgrad_products)(n, PP(*[dev_ptr(d, "delta") for d in deltas]), II(*[d.stride(0) for d in deltas]), II(*Ms),
                                           PP(*[dev_ptr(x, "x") for x in xs]), II(*[x.stride(0) for x in xs]), II(*Ns), int(P),
                                           PP(*[dev_ptr(o) for o in outs]), II(*Ns), PP(*[dev_ptr(b) for b in biases]) if want_bias else None,
                                           dev_ptr(scratch, "scratch", torch.uint8, 16), scratch.numel(), int(iters),
                                           C.byref(ms) if timed else None, stream_ptr(dev)), "mi_nerf_wgrad_products")
    return outs, biases, (float(ms.value) if timed else None)


def backward_range(net: Net, n_rays: int, S: int, work: torch.Tensor):
    """What the last split-precision backward over ``work`` saw (one device -> host read): (max |d_raw|, max |delta * s|), s the power of two
    that put max|d_raw| in [2^7, 2^8).  The second number is how much of the f16 range the scaled chain used: at or beyond 65504 a conversion
    saturated (FP16_OVFL: never inf) and the gradients of that step are clipped there."""
    lay = train_layout(net, n_rays, S)
    words = work[lay.work_bytes - 256:lay.work_bytes - 248].view(torch.float32).cpu()
    return float(words[0]), float(words[1])


def backward_range_words(net: Net, n_rays: int, S: int, work: torch.Tensor) -> torch.Tensor:
    """The same two words as a device view (float32 [2]) -- for callers that fold them on the device instead of reading them (train_path)."""
    lay = train_layout(net, n_rays, S)
    return work[lay.work_bytes - 256:lay.work_bytes - 248].view(torch.float32)


def train_views(net: Net, n_rays: int, S: int, stash: Optional[torch.Tensor] = None, work: Optional[torch.Tensor] = None):
    """Named float views into the stash / backward workspace (staged parity checks)."""
    lay = train_layout(net, n_rays, S)
    n_pts = n_rays * S
    W, D = net.W, net.D
    out = {}
    def view(buf, off, shape):
        cnt = int(np.prod(shape))
        return buf[off:off + 4 * cnt].view(torch.float32).reshape(shape)
    if stash is not None:
        out["stash_h"] = view(stash, lay.stash_h, (D, n_pts, W))
        out["stash_f"] = view(stash, lay.stash_f, (n_pts, W))
        out["stash_g"] = view(stash, lay.stash_g, (n_pts, W // 2))
    if work is not None:
        out["delta_h"] = view(work, lay.delta_h, (D, n_pts, W))
        out["delta_f"] = view(work, lay.delta_f, (n_pts, W))
        out["delta_d"] = view(work, lay.delta_d, (n_pts, W // 2))
    return out


# ------------------------------------------------------------------------------------------------
# either side of the path: frame metrics, 8-bit conversion, global-batch staging
# ------------------------------------------------------------------------------------------------
REDUCE_SCRATCH_BYTES = 8192


def image_metrics(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """[mse, psnr] (device tensor of 2 floats): img2mse + mse2psnr, utils.py:18-23."""
    if pred.shape != target.shape or pred.numel() == 0:
        raise MiNerfError(f"pred {tuple(pred.shape)} and target {tuple(target.shape)} must match and be non-empty")
    dev = pred.device
    out = torch.empty(2, dtype=torch.float32, device=dev)
    scratch = torch.empty(REDUCE_SCRATCH_BYTES, dtype=torch.uint8, device=dev)
    with _guard(dev):
        check(lib().mi_nerf_image_metrics(dev_ptr(pred, "pred"), dev_ptr(target, "target"), pred.numel(), dev_ptr(out),
                                          dev_ptr(scratch, "scratch", torch.uint8), scratch.numel(), stream_ptr(dev)), "mi_nerf_image_metrics")
    return out


def nanmax(x: torch.Tensor) -> torch.Tensor:
    if x.numel() == 0:
        raise MiNerfError("nanmax of an empty tensor")
    dev = x.device
    out = torch.empty(1, dtype=torch.float32, device=dev)
    scratch = torch.empty(REDUCE_SCRATCH_BYTES, dtype=torch.uint8, device=dev)
    with _guard(dev):
        check(lib().mi_nerf_nanmax(dev_ptr(x, "x"), x.numel(), dev_ptr(out), dev_ptr(scratch, "scratch", torch.uint8), scratch.numel(),
                                   stream_ptr(dev)), "mi_nerf_nanmax")
    return out


def to8b(x: torch.Tensor, divisor: Optional[torch.Tensor] = None) -> torch.Tensor:
    """uint8 image of x (or x / divisor[0]): utils.py:15 on the device."""
    dev = x.device
    out = torch.empty(x.shape, dtype=torch.uint8, device=dev)
    with _guard(dev):
        check(lib().mi_nerf_to8b(dev_ptr(x, "x"), x.numel(), dev_ptr(divisor, "divisor"), dev_ptr(out, "out", torch.uint8, 1), stream_ptr(dev)),
              "mi_nerf_to8b")
    return out


def rays_rgb(W: int, H: int, K, poses: torch.Tensor, images: torch.Tensor) -> torch.Tensor:
    """[n_img*H*W, 3, 3] (origin, direction, pixel) for every pixel of every image: main.py:92-101 in one launch."""
    n_img = poses.shape[0]
    if tuple(poses.shape[1:]) != (3, 4) or tuple(images.shape) != (n_img, H, W, 3):
        raise MiNerfError(f"poses must be [n,3,4] and images [n,{H},{W},3]; got {tuple(poses.shape)} / {tuple(images.shape)}")
    dev = images.device
    k = K.detach().cpu().numpy() if isinstance(K, torch.Tensor) else np.asarray(K)
    k4 = (C.c_float * 4)(float(np.float32(k[0][0])), float(np.float32(k[1][1])), float(np.float32(k[0][2])), float(np.float32(k[1][2])))
    out = torch.empty(n_img * H * W, 3, 3, dtype=torch.float32, device=dev)
    with _guard(dev):
        check(lib().mi_nerf_rays_rgb(int(W), int(H), k4, dev_ptr(poses, "poses"), dev_ptr(images, "images"), n_img, dev_ptr(out),
                                     stream_ptr(dev)), "mi_nerf_rays_rgb")
    return out


def gather_rows(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """``src[idx]`` along the leading axis (mi_nerf_permute_rows with n = len(idx)): the rows one training step consumes, gathered straight
    from the unshuffled table.  ``idx`` int64 on the device, every value < len(src) (the caller's cursor guarantees it; not checked here)."""
    if idx.dim() != 1 or idx.dtype != torch.int64:
        raise MiNerfError(f"idx must be a 1-d int64 tensor, got {tuple(idx.shape)} {idx.dtype}")
    n_src = src.shape[0]
    row = src.numel() // max(n_src, 1)
    n = int(idx.shape[0])
    dst = torch.empty((n, *src.shape[1:]), dtype=torch.float32, device=src.device)
    if n == 0:
        return dst
    with _guard(src.device):
        check(lib().mi_nerf_permute_rows(dev_ptr(src, "src"), dev_ptr(idx, "idx", torch.int64, 8), n, int(row), dev_ptr(dst), stream_ptr(src.device)),
              "mi_nerf_permute_rows")
    return dst


def permute_rows(src: torch.Tensor, perm: torch.Tensor) -> torch.Tensor:
    """``src[perm]`` for a permutation of ALL rows (np.random.shuffle of the leading axis, main.py:102)."""
    n = src.shape[0]
    if perm.shape != (n,):
        raise MiNerfError(f"perm must be [{n}], got {tuple(perm.shape)}")
    return gather_rows(src, perm)


# ------------------------------------------------------------------------------------------------
# fused render
# ------------------------------------------------------------------------------------------------
# MI_NERF_MODE_* of include/mi_nerf.h (mi_nerf_render_cfg.mode, mi_nerf_time_mlp_rays)
MODE_F32, MODE_BF16, MODE_BF16_64, MODE_BF16_32, MODE_F16S, MODE_F16S_BF16 = 0, 1, 2, 3, 5, 6
BF16_SHAPES = {0: MODE_BF16, 64: MODE_BF16_64, 32: MODE_BF16_32, 832: 4}          # points per wave -> mode (832: a retired shape the library refuses)


def render_cfg(near: float, far: float, Sc: int, Nf: int, det: bool, bf16: bool = False, points_per_wave: int = 0, seed: int = 0,
               ray_offset: int = 0, f16s: bool = False, coarse_f16s: bool = False) -> RenderCfg:
    """``points_per_wave`` (bf16 only): 0 = the bf16 kernel's launch shape is chosen per launch; 64 / 32 pin it.
    ``f16s``: the split-precision MLP variant (blobs from PackedNeRF.f16s()).
    ``bf16`` with ``coarse_f16s``: MI_NERF_MODE_F16S_BF16 -- the coarse network in split precision (fp32-grade fine sample positions),
    the fine network in bf16; the caller hands PackedNeRF.f16s()[0] and PackedNeRF.bf16()[1].
    ``seed`` / ``ray_offset`` key the jitter the kernels draw themselves when render_rays gets no ``t_rand`` / ``u`` tensor."""
    if bf16 and f16s:
        raise MiNerfError("bf16 and f16s are different precision modes: pick one (bf16 with coarse_f16s=True mixes them per network)")
    if coarse_f16s and (not bf16 or points_per_wave):
        raise MiNerfError("coarse_f16s goes with bf16=True (fine network in bf16, launch shape chosen per launch)")
    mode = MODE_F16S if f16s else (MODE_F16S_BF16 if coarse_f16s else (BF16_SHAPES[int(points_per_wave)] if bf16 else MODE_F32))
    return RenderCfg(float(near), float(far), int(Sc), int(Nf), int(bool(det)), mode, int(seed) & 0xFFFFFFFF, 0, int(ray_offset))


def workspace_layout(cfg: RenderCfg, n: int) -> WorkspaceLayout:
    wl = WorkspaceLayout()
    check(lib().mi_nerf_render_workspace_layout(C.byref(cfg), n, C.byref(wl)), "mi_nerf_render_workspace_layout")
    return wl


def render_rays(net: Net, packed_c: torch.Tensor, packed_f: Optional[torch.Tensor], cfg: RenderCfg, rays: torch.Tensor,
                t_rand: Optional[torch.Tensor], u: Optional[torch.Tensor], workspace: Optional[torch.Tensor] = None,
                out: Optional[Sequence[torch.Tensor]] = None):
    """One mi_nerf_render_rays call.  Returns (rgb_c, disp_c, rgb_f|None, disp_f|None, workspace).
    ``t_rand`` / ``u`` None: the kernels draw the jitter themselves from (cfg.seed, cfg.ray_offset + ray, sample)."""
    n = rays.shape[0]
    dev = rays.device
    if tuple(rays.shape) != (n, 6) or (t_rand is not None and tuple(t_rand.shape) != (n, cfg.Sc)):
        raise MiNerfError(f"rays [n,6] / t_rand [n,{cfg.Sc}] expected, got {tuple(rays.shape)} / {None if t_rand is None else tuple(t_rand.shape)}")
    if cfg.Nf > 0 and not cfg.det and u is not None and tuple(u.shape) != (n, cfg.Nf):
        raise MiNerfError(f"u [n,{cfg.Nf}] expected")
    wl = workspace_layout(cfg, n)
    if workspace is None or workspace.numel() < wl.total:
        workspace = torch.empty(max(wl.total, 256), dtype=torch.uint8, device=dev)
    if out is None:
        rgb_c = torch.empty(n, 3, dtype=torch.float32, device=dev)
        disp_c = torch.empty(n, dtype=torch.float32, device=dev)
        rgb_f = torch.empty(n, 3, dtype=torch.float32, device=dev) if cfg.Nf > 0 else None
        disp_f = torch.empty(n, dtype=torch.float32, device=dev) if cfg.Nf > 0 else None
    else:
        rgb_c, disp_c, rgb_f, disp_f = out
    with _guard(dev):
        check(lib().mi_nerf_render_rays(C.byref(net), dev_ptr(packed_c, "packed_coarse", torch.uint8, 16),
                                        dev_ptr(packed_f, "packed_fine", torch.uint8, 16), C.byref(cfg), dev_ptr(rays, "rays"), n,
                                        dev_ptr(t_rand, "t_rand"), dev_ptr(u, "u") if (cfg.Nf > 0 and not cfg.det) else None,
                                        dev_ptr(workspace, "workspace", torch.uint8, 256), workspace.numel(), dev_ptr(rgb_c), dev_ptr(disp_c),
                                        dev_ptr(rgb_f), dev_ptr(disp_f), stream_ptr(dev)), "mi_nerf_render_rays")
    return rgb_c, disp_c, rgb_f, disp_f, workspace


def workspace_views(cfg: RenderCfg, n: int, workspace: torch.Tensor) -> Dict[str, torch.Tensor]:
    """Typed views of the intermediates inside a render workspace (staged parity checks)."""
    wl = workspace_layout(cfg, n)
    Sc, St = cfg.Sc, cfg.Sc + cfg.Nf
    def view(off, shape):
        cnt = int(np.prod(shape))
        return workspace[off:off + cnt * 4].view(torch.float32).view(*shape)
    v = {"z_c": view(wl.z_c, (n, Sc)), "raw_c": view(wl.raw_c, (n, Sc, 4)), "weights_c": view(wl.weights_c, (n, Sc))}
    if cfg.Nf > 0:
        v["z_f"] = view(wl.z_f, (n, St))
        v["raw_f"] = view(wl.raw_f, (n, St, 4))
    return v


def time_mlp_rays(net: Net, packed: torch.Tensor, rays: torch.Tensor, z: torch.Tensor, raw: torch.Tensor, iters: int, bf16: bool = False,
                  points_per_wave: int = 0, f16s: bool = False) -> float:
    """Average device milliseconds per fused-MLP launch, from hipEvents on the launch stream."""
    n, S = z.shape
    ms = C.c_float(0.0)
    with _guard(z.device):
        check(lib().mi_nerf_time_mlp_rays(C.byref(net), dev_ptr(packed, "packed", torch.uint8, 16), dev_ptr(rays, "rays"), dev_ptr(z, "z"), n, S,
                                          dev_ptr(raw, "raw", align=16), iters, MODE_F16S if f16s else (BF16_SHAPES[int(points_per_wave)] if bf16 else 0), C.byref(ms),
                                          stream_ptr(z.device)), "mi_nerf_time_mlp_rays")
    return float(ms.value)


def selftest_mfma(device) -> None:
    device = torch.device(device)
    with _guard(device):
        check(lib().mi_nerf_selftest_mfma(stream_ptr(device)), "mi_nerf_selftest_mfma")
