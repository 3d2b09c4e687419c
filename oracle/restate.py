"""CPU restatement of the reference's volume-rendering hot path.  TEST INFRASTRUCTURE ONLY.

This module is the parity oracle for the HIP path.  It may be imported only from
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg -- never from
the product package ``nerf_pytorch_paeng_amd`` (which has no CPU fallback and fails loudly
when ``libmi_nerf.so`` is missing).

Parity status: PINNED.  Every function below is checked against tensors captured from the
reference itself (fixtures F1..F10 under ``tests/golden/``, produced by
``oracle/gen_fixtures.py`` which imports ``/root/reference`` in the build container) and
against the hand-computed known-answer vectors of SURVEY.md section 8(a).  The training path is checked against
torch autograd run on these same functions (``fine_z`` detaches like nerf_process.py:66).

All arithmetic is fp32 (torch CPU).  Randomness is always *injected* (``t_rand`` for the
stratified jitter, ``u`` for the inverse-CDF draw) because the reference draws unseeded
``torch.rand`` (nerf_process.py:58-60, :162-163) and is therefore not reproducible by itself.

Citations are ``file:line`` into the reference tree.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

F32 = torch.float32


# --------------------------------------------------------------------------------------------
# configuration bag (the subset of the reference's ``opts`` Namespace the path reads;
# config.py:35-36,54-57,72-76 and SURVEY.md section 5)
# --------------------------------------------------------------------------------------------
@dataclass
class PathConfig:
    near: float = 2.0
    far: float = 6.0
    N_samples_c: int = 64
    N_samples_f: int = 128
    perturb: float = 1.0          # det = (perturb == 0.)         nerf_process.py:65
    chunk_rays: int = 4096
    chunk_pts: int = 524288
    data_type: str = "blender"    # 'llff' switches the NDC warp on  nerf_process.py:224
    L_x: int = 10
    L_d: int = 4
    netDepth: int = 8
    netWidth: int = 256
    skips: Tuple[int, ...] = (4,)


# --------------------------------------------------------------------------------------------
# a1 / a2  ray generation                                                  rays.py:7-34
# --------------------------------------------------------------------------------------------
def make_o_d(img_w: int, img_h: int, img_k, pose) -> Tuple[torch.Tensor, torch.Tensor]:
    """Pinhole rays for every pixel (rays.py:20-34).

    Pixel (row y, column x) gets camera-frame direction ((x-cx)/fx, -(y-cy)/fy, -1)
    (rays.py:28-30, no half-pixel offset), rotated by the 3x3 block of ``pose`` (rays.py:32);
    the origin is the translation column broadcast to every pixel (rays.py:33).
    ``img_k`` may be float64 (numpy or tensor): torch type promotion keeps the result fp32
    because the pixel grid is an fp32 *tensor* and K entries are 0-dim operands.
    """
    pose = torch.as_tensor(pose, dtype=F32)
    k = np.asarray(torch.as_tensor(img_k).cpu().numpy(), dtype=np.float64)
    xs = torch.arange(img_w, dtype=F32)[None, :].expand(img_h, img_w)
    ys = torch.arange(img_h, dtype=F32)[:, None].expand(img_h, img_w)
    # 0-dim float64 scalars do not promote an fp32 tensor: arithmetic below is fp32, with the
    # scalar first rounded to fp32 exactly as torch does for python/0-dim operands.
    cx, cy = float(np.float32(k[0, 2])), float(np.float32(k[1, 2]))
    fx, fy = float(np.float32(k[0, 0])), float(np.float32(k[1, 1]))
    dx = (xs - cx) / fx
    dy = -(ys - cy) / fy
    dz = -torch.ones_like(dx)
    dirs = torch.stack([dx, dy, dz], dim=-1)                    # [H, W, 3]
    rot = pose[:3, :3]
    rays_d = dirs @ rot.T                                        # rays.py:32
    rays_o = pose[:3, 3].expand(rays_d.shape)                    # rays.py:33
    return rays_o, rays_d


def get_rays_np(H: int, W: int, K, c2w) -> Tuple[np.ndarray, np.ndarray]:
    """numpy twin used for global-batch precompute (rays.py:7-17); dtype follows K."""
    K = np.asarray(K)
    c2w = np.asarray(c2w)
    xs, ys = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing="xy")
    cam = np.stack([(xs - K[0][2]) / K[0][0], -(ys - K[1][2]) / K[1][1], -np.ones_like(xs)], axis=-1)
    rays_d = (cam[..., None, :] * c2w[:3, :3]).sum(-1)          # rays.py:14
    rays_o = np.broadcast_to(c2w[:3, -1], rays_d.shape)          # rays.py:16
    return rays_o, rays_d


# --------------------------------------------------------------------------------------------
# a3  NDC warp for forward-facing (LLFF) scenes                       nerf_process.py:8-28
# --------------------------------------------------------------------------------------------
def ndc_rays(H: int, W: int, focal: float, near: float, rays_o: torch.Tensor, rays_d: torch.Tensor):
    focal = float(focal)
    # nerf_process.py:11-12  slide the origin onto the z = -near plane
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    o = rays_o + t[..., None] * rays_d
    # nerf_process.py:15-23; the reference writes the scale as -1/(W/(2 f)) -- python-float
    # arithmetic (float64) rounded once to fp32 when it meets the tensor.
    sx = -1.0 / (W / (2.0 * focal))
    sy = -1.0 / (H / (2.0 * focal))
    ox_oz = o[..., 0] / o[..., 2]
    oy_oz = o[..., 1] / o[..., 2]
    o0 = sx * o[..., 0] / o[..., 2]
    o1 = sy * o[..., 1] / o[..., 2]
    o2 = 1.0 + 2.0 * near / o[..., 2]
    d0 = sx * (rays_d[..., 0] / rays_d[..., 2] - ox_oz)
    d1 = sy * (rays_d[..., 1] / rays_d[..., 2] - oy_oz)
    d2 = -2.0 * near / o[..., 2]
    return torch.stack([o0, o1, o2], -1), torch.stack([d0, d1, d2], -1)


# --------------------------------------------------------------------------------------------
# a6  stratified coarse sampling                                     nerf_process.py:42-60
# --------------------------------------------------------------------------------------------
def stratified_z(n_rays: int, near: float, far: float, n_samples: int, t_rand: torch.Tensor) -> torch.Tensor:
    """z = lower + (upper-lower) * t_rand over ``n_samples`` depth bins, linear in depth.

    The jitter is unconditional in the reference (nerf_process.py:58-60: no ``perturb`` test).
    """
    t = torch.linspace(0.0, 1.0, steps=n_samples, dtype=F32)
    near_col = near * torch.ones(n_rays, 1, dtype=F32)
    far_col = far * torch.ones(n_rays, 1, dtype=F32)
    z = near_col * (1.0 - t) + far_col * t                       # :53
    z = z.expand(n_rays, n_samples)
    mids = 0.5 * (z[..., 1:] + z[..., :-1])                      # :55
    upper = torch.cat([mids, z[..., -1:]], -1)                   # :56
    lower = torch.cat([z[..., :1], mids], -1)                    # :57
    return lower + (upper - lower) * t_rand.to(F32)              # :60


# --------------------------------------------------------------------------------------------
# a7  hierarchical inverse-CDF sampling                         nerf_process.py:62-67,144-182
# --------------------------------------------------------------------------------------------
def sample_pdf(bins: torch.Tensor, weights: torch.Tensor, n_samples: int, det: bool,
               u: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Inverse-transform samples of the piecewise-constant pdf ``weights`` over ``bins``.

    ``bins`` [n, B], ``weights`` [n, B-1].  ``u`` must be given unless ``det``.
    """
    w = weights + 1e-5                                            # :150
    pdf = w / torch.sum(w, -1, keepdim=True)                      # :151
    cdf = torch.cumsum(pdf, -1)                                   # :152
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)    # :154  -> [n, B]
    if det:
        u = torch.linspace(0.0, 1.0, steps=n_samples, dtype=F32).expand(list(cdf.shape[:-1]) + [n_samples])  # :158-160
    else:
        assert u is not None, "inject u (the reference draws torch.rand here, nerf_process.py:162)"
        u = u.to(F32)
    u = u.contiguous()
    idx = torch.searchsorted(cdf, u, right=True)                  # :167   #{cdf <= u}
    lo = (idx - 1).clamp(min=0)                                   # :168
    hi = idx.clamp(max=cdf.shape[-1] - 1)                         # :169
    cdf_lo, cdf_hi = torch.gather(cdf, -1, lo), torch.gather(cdf, -1, hi)     # :174-175
    bin_lo, bin_hi = torch.gather(bins, -1, lo), torch.gather(bins, -1, hi)   # :176
    denom = cdf_hi - cdf_lo                                       # :178
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)          # :179
    t = (u - cdf_lo) / denom                                      # :180
    return bin_lo + t * (bin_hi - bin_lo)                         # :181


def fine_z(z_coarse: torch.Tensor, weights_coarse: torch.Tensor, n_fine: int, det: bool,
           u: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Merged, sorted fine sample depths (nerf_process.py:63-67).  Returns (z_fine, z_samples)."""
    mids = 0.5 * (z_coarse[..., 1:] + z_coarse[..., :-1])         # :63
    z_new = sample_pdf(mids, weights_coarse[..., 1:-1], n_fine, det, u)       # :64-65
    z_new = z_new.detach()                                        # :66 -- the fine loss never reaches the coarse network
    z_all, _ = torch.sort(torch.cat([z_coarse, z_new], -1), -1)   # :67
    return z_all, z_new


# --------------------------------------------------------------------------------------------
# a8  positional encoding and network-input assembly
#     model/PositionalEncoding.py:7-36, nerf_process.py:36-39,69-85
# --------------------------------------------------------------------------------------------
def posenc(x: torch.Tensor, L: int) -> torch.Tensor:
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)], 3-wide blocks."""
    bands = 2.0 ** torch.linspace(0.0, L - 1, L)                  # PositionalEncoding.py:18
    out = [x]
    for f in bands:
        out.append(torch.sin(x * f))
        out.append(torch.cos(x * f))
    return torch.cat(out, -1)


def embed(rays: torch.Tensor, z_vals: torch.Tensor, L_x: int, L_d: int) -> torch.Tensor:
    """[n*S, (3+6 L_x) + (3+6 L_d)] network input.  View directions are the *given* ray
    directions normalised (nerf_process.py:37-39) -- NDC-space directions for llff."""
    o, d = rays[:, :3], rays[:, 3:]
    view = d / torch.norm(d, dim=-1, keepdim=True)
    pts = o[:, None, :] + d[:, None, :] * z_vals[..., None]       # :69-70
    n, s = z_vals.shape
    gx = posenc(pts.reshape(-1, 3), L_x)                          # :73
    gd = posenc(view[:, None, :].expand(n, s, 3).reshape(-1, 3), L_d)         # :77-81
    return torch.cat([gx, gd], -1)                                # :83-84


# --------------------------------------------------------------------------------------------
# a9  the MLP                                                      model/NeRF.py:10-52
# --------------------------------------------------------------------------------------------
def mlp_forward(sd: Dict[str, torch.Tensor], prefix: str, x: torch.Tensor, D: int,
                in_x: int, in_d: int, skips: Sequence[int] = (4,), dtype=F32, taps: Optional[dict] = None) -> torch.Tensor:
    """One NeRFModule forward from a reference-layout ``state_dict`` (keys
    ``{prefix}linear_x.{i}.weight`` ..., NeRF.py:24-30).  Returns [n, 4] = (rgb_raw, density_raw).
    ``taps`` (tests of the training path): collects the pre-activations ``a{i}``, ``feat``, ``ad`` (with
    retain_grad when they carry a graph) so per-layer gradients can be compared."""
    def tap(name, v):
        if taps is not None:
            if v.requires_grad:
                v.retain_grad()
            taps[name] = v
        return v
    def lin(name, v):
        w = torch.as_tensor(sd[f"{prefix}{name}.weight"]).to(dtype)
        b = torch.as_tensor(sd[f"{prefix}{name}.bias"]).to(dtype)
        return v @ w.T + b
    x = x.to(dtype)
    gx, gd = x[:, :in_x], x[:, in_x:in_x + in_d]                  # NeRF.py:34
    h = gx
    for i in range(D):                                            # NeRF.py:37-41
        h = torch.relu(tap(f"a{i}", lin(f"linear_x.{i}", h)))
        if i in skips:
            h = torch.cat([gx, h], -1)                            # order: [input_x, out]
    sigma = lin("linear_density", h)                              # NeRF.py:43
    feat = tap("feat", lin("linear_feat", h))                     # NeRF.py:44 (no activation)
    h = torch.relu(tap("ad", lin("linear_d", torch.cat([feat, gd], -1))))    # NeRF.py:46-48
    rgb = lin("linear_color", h)                                  # NeRF.py:50
    return torch.cat([rgb, sigma], -1)                            # NeRF.py:51


def bf16_round(t: torch.Tensor) -> torch.Tensor:
    """Round to bfloat16 (nearest even) and return in the input's dtype."""
    return t.to(torch.float32).to(torch.bfloat16).to(t.dtype)


def mlp_forward_bf16(sd: Dict[str, torch.Tensor], prefix: str, x: torch.Tensor, D: int, in_x: int, in_d: int,
                     skips: Sequence[int] = (4,), dtype=F32) -> torch.Tensor:
    """``mlp_forward`` with the ROUNDING POINTS of the bf16 MFMA variant (BASELINE config #5; nerf_pytorch_paeng_amd/csrc/mlp_bf16.hip):
    weights of every Linear rounded to bf16 (except the view-direction columns of linear_d, which the kernel folds into a per-ray
    fp32 bias), gamma(x) rounded to bf16, every activation rounded to bf16 where it becomes the next layer's input (after ReLU;
    linear_feat's output without ReLU), products accumulated in ``dtype`` (fp32 on the device), biases in full precision.  The
    network is model/NeRF.py:33-52 unchanged.  This is the oracle the bf16 kernel is checked against; PSNR against the fp32 path
    is reported beside it."""
    def wq(name, cols=None):
        w = torch.as_tensor(sd[f"{prefix}{name}.weight"]).float()
        if cols is None:
            return bf16_round(w).to(dtype)
        w = w.clone()
        w[:, cols] = bf16_round(w[:, cols])
        return w.to(dtype)
    def b(name):
        return torch.as_tensor(sd[f"{prefix}{name}.bias"]).to(dtype)
    x = x.to(dtype)
    gx, gd = bf16_round(x[:, :in_x]), x[:, in_x:in_x + in_d]
    h = gx
    for i in range(D):
        h = bf16_round(torch.relu(h @ wq(f"linear_x.{i}").T + b(f"linear_x.{i}")))
        if i in skips:
            h = torch.cat([gx, h], -1)
    W = torch.as_tensor(sd[f"{prefix}linear_feat.weight"]).shape[0]
    sigma = h @ wq("linear_density").T + b("linear_density")
    feat = bf16_round(h @ wq("linear_feat").T + b("linear_feat"))
    g = bf16_round(torch.relu(torch.cat([feat, gd], -1) @ wq("linear_d", slice(0, W)).T + b("linear_d")))
    rgb = g @ wq("linear_color").T + b("linear_color")
    return torch.cat([rgb, sigma], -1)


def f16_split(t: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """t = hi + lo with hi = f16(t) and lo = f16((t - hi) * 2^11) * 2^-11, both returned as fp32 (the operand format of the
    split-precision MFMA variant, nerf_pytorch_paeng_amd/csrc/mlp_f16s.hip)."""
    t = t.to(F32)
    hi = t.to(torch.float16).to(F32)
    lo = ((t - hi) * 2048.0).to(torch.float16).to(F32) / 2048.0
    return hi, lo


def mlp_forward_f16split(sd: Dict[str, torch.Tensor], prefix: str, x: torch.Tensor, D: int, in_x: int, in_d: int,
                         skips: Sequence[int] = (4,), dtype=torch.float64) -> torch.Tensor:
    """``mlp_forward`` with the ROUNDING POINTS of the split-precision variant (mlp_f16s.hip): every weight, every encoded input and
    every activation (after ReLU; linear_feat's output without) is carried as an f16 pair hi + lo, a product is hi.hi + hi.lo + lo.hi
    (lo.lo dropped); the view-direction columns of linear_d are folded into a per-ray fp32 bias and not split.  Partial products are
    accumulated in ``dtype`` (fp64 by default: what is left against the kernel is its fp32 summation order).  The network is
    model/NeRF.py:33-52 unchanged; the result is expected to be as close to an fp64 evaluation as the reference's own fp32 arithmetic."""
    def split_lin(name, v, cols=None):
        w = torch.as_tensor(sd[f"{prefix}{name}.weight"]).float()
        b = torch.as_tensor(sd[f"{prefix}{name}.bias"]).to(dtype)
        if cols is not None:
            w_rest = torch.cat([w[:, :cols.start], w[:, cols.stop:]], -1).to(dtype)
            w = w[:, cols]
        wh, wl = (t.to(dtype) for t in f16_split(w))
        vs = v if cols is None else v[:, cols]
        vh, vl = (t.to(dtype) for t in f16_split(vs))
        out = vh @ wh.T + (vl @ wh.T + vh @ wl.T) + b
        if cols is not None:
            out = out + torch.cat([v[:, :cols.start], v[:, cols.stop:]], -1).to(dtype) @ w_rest.T
        return out.to(F32)                       # the kernel combines acc_hi + acc_lo * 2^-11 in fp32 before it splits again
    x = x.to(F32)
    gx, gd = x[:, :in_x], x[:, in_x:in_x + in_d]
    h = gx
    for i in range(D):
        h = torch.relu(split_lin(f"linear_x.{i}", h))
        if i in skips:
            h = torch.cat([gx, h], -1)
    W = torch.as_tensor(sd[f"{prefix}linear_feat.weight"]).shape[0]
    sigma = split_lin("linear_density", h)
    feat = split_lin("linear_feat", h)
    g = torch.relu(split_lin("linear_d", torch.cat([feat, gd], -1), slice(0, W)))
    rgb = split_lin("linear_color", g)
    return torch.cat([rgb, sigma], -1)


def run_network(sd, x: torch.Tensor, cfg: PathConfig, is_fine: bool, chunk: Optional[int] = None,
                dtype=F32) -> torch.Tensor:
    """Chunked evaluation (nerf_process.py:190-192,206-207; NeRF.py:70-78)."""
    prefix = "model_fine." if is_fine else "model_coarse."
    in_x, in_d = 3 + 6 * cfg.L_x, 3 + 6 * cfg.L_d
    chunk = chunk or cfg.chunk_pts
    outs = [mlp_forward(sd, prefix, x[i:i + chunk], cfg.netDepth, in_x, in_d, cfg.skips, dtype)
            for i in range(0, x.shape[0], chunk)]
    return torch.cat(outs, 0)


# --------------------------------------------------------------------------------------------
# a10  alpha compositing                                           nerf_process.py:89-140
# --------------------------------------------------------------------------------------------
def post_process(raw: torch.Tensor, z_vals: torch.Tensor, rays_d: torch.Tensor):
    """raw [n,S,4], z [n,S], rays_d [n,3] -> (rgb_map, disp_map, acc_map, weights, depth_map)."""
    dists = z_vals[..., 1:] - z_vals[..., :-1]                    # :93
    dists = torch.cat([dists, torch.full_like(dists[..., :1], 1e10)], -1)     # :96-97
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)      # :101
    rgb = torch.sigmoid(raw[..., :3])                             # :103-104
    alpha = 1.0 - torch.exp(-torch.relu(raw[..., 3]) * dists)     # :91-92,106 (no density noise)
    ones = torch.ones(alpha.shape[0], 1, dtype=alpha.dtype)
    trans = torch.cumprod(torch.cat([ones, 1.0 - alpha + 1e-10], -1), -1)[:, :-1]   # :109-110
    weights = alpha * trans                                       # :111
    rgb_map = torch.sum(weights[..., None] * rgb, -2)             # :113
    depth_map = torch.sum(weights * z_vals, -1)                   # :114
    acc_map = torch.sum(weights, -1)                              # :136
    disp = 1.0 / torch.maximum(1e-10 * torch.ones_like(depth_map), depth_map / acc_map)   # :124-125
    disp = torch.where(torch.isnan(disp), torch.zeros_like(disp), disp)       # :126-127
    disp = torch.where(disp > 5.0, 5.0 * torch.ones_like(disp), disp)         # :132-134
    rgb_map = rgb_map + (1.0 - acc_map[..., None])                # :138  white bg, always
    return rgb_map, disp, acc_map, weights, depth_map


# --------------------------------------------------------------------------------------------
# a5  render_rays, a4 batchify                              nerf_process.py:185-216,220-252
# --------------------------------------------------------------------------------------------
def render_rays(rays: torch.Tensor, sd, cfg: PathConfig, t_rand: torch.Tensor,
                u: Optional[torch.Tensor] = None, z_fine_override: Optional[torch.Tensor] = None,
                mlp_dtype=F32) -> Dict[str, torch.Tensor]:
    """Coarse pass -> composite -> resample -> fine pass over all N_c+N_f sorted depths.

    Returns the reference's output dict plus every intermediate (keys prefixed ``_``).
    ``z_fine_override`` pins the fine sample positions (staged parity: SURVEY.md section 7,
    "sample_pdf is discontinuous").
    """
    n = rays.shape[0]
    rays = rays.to(F32)
    z_c = stratified_z(n, cfg.near, cfg.far, cfg.N_samples_c, t_rand)
    emb_c = embed(rays, z_c, cfg.L_x, cfg.L_d)
    raw_c = run_network(sd, emb_c, cfg, False, dtype=mlp_dtype).to(F32).reshape(n, cfg.N_samples_c, 4)
    rgb_c, disp_c, acc_c, w_c, depth_c = post_process(raw_c, z_c, rays[:, 3:])
    out = {"rgb_c": rgb_c, "disp_c": disp_c, "_z_c": z_c, "_raw_c": raw_c, "_weights_c": w_c,
           "_acc_c": acc_c, "_depth_c": depth_c}
    if cfg.N_samples_f > 0:
        det = (cfg.perturb == 0.0)
        z_f, z_new = fine_z(z_c, w_c, cfg.N_samples_f, det, u)
        if z_fine_override is not None:
            z_f = z_fine_override.to(F32)
        emb_f = embed(rays, z_f, cfg.L_x, cfg.L_d)
        raw_f = run_network(sd, emb_f, cfg, True, dtype=mlp_dtype).to(F32).reshape(n, z_f.shape[1], 4)
        rgb_f, disp_f, acc_f, w_f, depth_f = post_process(raw_f, z_f, rays[:, 3:])
        out.update({"rgb_f": rgb_f, "disp_f": disp_f, "_z_f": z_f, "_z_samples": z_new, "_raw_f": raw_f,
                    "_weights_f": w_f, "_acc_f": acc_f, "_depth_f": depth_f})
    return out


def batchify_rays_and_render_by_chunk(ray_o, ray_d, sd, H: int, W: int, K, cfg: PathConfig,
                                      t_rand: torch.Tensor, u: Optional[torch.Tensor] = None):
    """Flatten, optional NDC, chunk by ``chunk_rays`` (nerf_process.py:220-252).

    ``t_rand`` [N, N_c] / ``u`` [N, N_f] are indexed by global ray so the result is
    independent of the chunk size (the reference re-draws per chunk)."""
    o = torch.as_tensor(ray_o, dtype=F32).reshape(-1, 3)
    d = torch.as_tensor(ray_d, dtype=F32).reshape(-1, 3)
    if cfg.data_type == "llff":
        o, d = ndc_rays(H, W, float(np.asarray(K)[0][0]), 1.0, o, d)          # :224-226
    rays = torch.cat([o, d], -1)                                  # :229
    parts = []
    for i in range(0, rays.shape[0], cfg.chunk_rays):             # :236
        uu = None if u is None else u[i:i + cfg.chunk_rays]
        parts.append(render_rays(rays[i:i + cfg.chunk_rays], sd, cfg, t_rand[i:i + cfg.chunk_rays], uu))
    rgb_c = torch.cat([p["rgb_c"] for p in parts], 0)
    disp_c = torch.cat([p["disp_c"] for p in parts], 0)
    if cfg.N_samples_f > 0:
        return rgb_c, disp_c, torch.cat([p["rgb_f"] for p in parts], 0), torch.cat([p["disp_f"] for p in parts], 0)
    return rgb_c, disp_c, None, None


# --------------------------------------------------------------------------------------------
# numpy mirror of the product's counter-based uniform generator (csrc/rng.h) -- lets tests and
# the CPU baseline use bit-identical t_rand / u without a GPU.
# --------------------------------------------------------------------------------------------
def _fmix32(h: np.ndarray) -> np.ndarray:
    h = h.astype(np.uint32, copy=True)
    h ^= h >> np.uint32(16)
    h *= np.uint32(0x85EBCA6B)
    h ^= h >> np.uint32(13)
    h *= np.uint32(0xC2B2AE35)
    h ^= h >> np.uint32(16)
    return h


def counter_uniform(seed: int, stream: int, ray0: int, n_rays: int, n_samples: int) -> np.ndarray:
    """U[0,1) fp32 [n_rays, n_samples]; value depends only on (seed, stream, global ray, sample)."""
    with np.errstate(over="ignore"):
        ray = (np.arange(n_rays, dtype=np.uint64) + np.uint64(ray0)).astype(np.uint32)[:, None]
        smp = np.arange(n_samples, dtype=np.uint32)[None, :]
        h = _fmix32(ray + np.uint32((0x9E3779B9 * (seed & 0xFFFFFFFF)) & 0xFFFFFFFF) + np.uint32(0x632BE5AB))
        k = smp * np.uint32(0x85EBCA6B) + np.uint32((stream * 0xC2B2AE35) & 0xFFFFFFFF) + np.uint32(0x27D4EB2F)
        h = _fmix32(h ^ k)
    return ((h >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)


def mse2psnr(mse: float) -> float:
    """utils.py:10-12."""
    return -10.0 * math.log(max(mse, 1e-30)) / math.log(10.0)


# --------------------------------------------------------------------------------------------
# callers either side of the path: metrics, 8-bit frames, global batch
# --------------------------------------------------------------------------------------------
def img2mse(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    return torch.mean((x - y) ** 2)                               # utils.py:18


def to8b(x: np.ndarray) -> np.ndarray:
    return (255 * np.clip(x, 0, 1)).astype(np.uint8)              # utils.py:15


def disp8(disp: np.ndarray) -> np.ndarray:
    return to8b(disp / np.nanmax(disp))                           # test.py:56


def rays_rgb_global_batch(H: int, W: int, K, poses: np.ndarray, images: np.ndarray, i_train) -> np.ndarray:
    """main.py:92-101: [len(i_train)*H*W, 3, 3] float32 = (origin, direction, pixel) per ray, unshuffled."""
    rays = np.stack([np.stack(get_rays_np(H, W, K, p), 0) for p in poses[:, :3, :4]], 0)          # [N, ro+rd, H, W, 3]
    rays_rgb = np.concatenate([rays, images[:, None]], 1)                                          # [N, 3, H, W, 3]
    rays_rgb = np.transpose(rays_rgb, [0, 2, 3, 1, 4])
    rays_rgb = np.stack([rays_rgb[i] for i in i_train], 0)
    return np.reshape(rays_rgb, [-1, 3, 3]).astype(np.float32)


def pose_spherical(theta: float, phi: float, radius: float) -> torch.Tensor:
    """dataset/render_pose.py:6-34."""
    trans_t = torch.Tensor(np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]])).float()
    ph, th = phi / 180. * np.pi, theta / 180. * np.pi
    rot_phi = torch.Tensor(np.array([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0], [0, np.sin(ph), np.cos(ph), 0], [0, 0, 0, 1]])).float()
    rot_theta = torch.Tensor(np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]])).float()
    c2w = rot_theta @ (rot_phi @ trans_t)
    return torch.Tensor(np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]])) @ c2w
