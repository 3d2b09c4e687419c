"""Volume-rendering hot path -- drop-in for the reference's nerf_process.py (252 lines).

Same function names, positional arguments and return structures as the reference
(``batchify_rays_and_render_by_chunk``, ``render_rays``, ``pre_process``, ``post_process``,
``sample_pdf``, ``ndc_rays``), executed by hand-written HIP kernels through ``libmi_nerf.so``.
Keyword-only extras (``t_rand=``, ``u=``, ``seed=``, ``ray_offset=``) make the randomness explicit:
the reference draws unseeded ``torch.rand`` (nerf_process.py:58-60,162-163); here the default is a
counter-based generator keyed on (seed, global ray index, sample index), so a frame renders
identically however its rays are chunked or sharded across GPUs.

Under ``torch.no_grad()`` (test.py:36,140) this is the forward-only inference path.  With gradients enabled
and a model whose parameters require them (train.py:53-70) the same entry points run the training path
(train_path.py): same kernels plus an activation stash, hand-written backward behind one autograd node.
Inputs are borrowed, outputs are fresh fp32 tensors on the inputs' device.
``opts`` fields read: near, far, N_samples_c, N_samples_f, perturb, chunk_rays, data_type
(gpu_ids / rank / chunk_pts are accepted and unused: the device comes from the tensors and the fused
kernel never materialises the [n_pts, 90] network input that chunk_pts exists to bound).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops
from ._lib import MiNerfError, as_f32_dev
from .weights import PackedNeRF, packed_for
from . import train_path

# rays handed to one mi_nerf_render_rays call (workspace: 5.4 KB/ray at 64+128 samples -> ~5.6 GB)
MAX_RAYS_PER_LAUNCH = 1 << 20

_rng = {"seed": 0, "calls": 0}


def manual_seed(seed: int) -> None:
    """Seed the default jitter generator (the reference is unseeded and irreproducible)."""
    _rng["seed"], _rng["calls"] = int(seed), 0


def _next_seed(seed: Optional[int]) -> int:
    if seed is not None:
        return int(seed)
    s = (_rng["seed"] * 0x9E3779B1 + _rng["calls"]) & 0xFFFFFFFF
    _rng["calls"] += 1
    return s


def _det(opts) -> bool:
    # the reference compares the raw attribute with 0. (nerf_process.py:65); a value read from a config
    # file can be a string (config.py:76 has no type=), for which `== 0.` is False
    p = getattr(opts, "perturb", 1.0)
    return isinstance(p, (int, float)) and p == 0.0


# --------------------------------------------------------------------------------------------------
def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """NDC warp for forward-facing scenes (nerf_process.py:8-28)."""
    lead = rays_d.shape[:-1]
    d = as_f32_dev(rays_d).reshape(-1, 3)
    o = rays_o.to(torch.float32) if rays_o.dtype != torch.float32 else rays_o
    o = o.expand(rays_d.shape).reshape(-1, 3) if o.shape != rays_d.shape else o.reshape(-1, 3)
    if isinstance(focal, torch.Tensor):
        focal = float(focal.item())
    oo, dd = ops.ndc_rays(int(H), int(W), float(focal), float(near), o, d)
    return oo.reshape(*lead, 3), dd.reshape(*lead, 3)


def sample_pdf(bins, weights, N_samples, det=False, opts=None, *, u=None):
    """Hierarchical inverse-CDF sampling (nerf_process.py:144-182).  ``u`` [n, N_samples] injects the
    uniforms; by default they come from the counter-based generator."""
    assert opts is not None                                         # nerf_process.py:147
    bins, weights = as_f32_dev(bins), as_f32_dev(weights, bins.device)
    lead = bins.shape[:-1]
    b2, w2 = bins.reshape(-1, bins.shape[-1]), weights.reshape(-1, weights.shape[-1])
    if not det:
        u = (ops.fill_uniform(_next_seed(None), 1, 0, b2.shape[0], int(N_samples), bins.device) if u is None
             else as_f32_dev(u, bins.device).reshape(-1, int(N_samples)))
    out = ops.sample_pdf(b2, w2, int(N_samples), bool(det), u)
    return out.reshape(*lead, int(N_samples))


def pre_process(rays, posenc, opts, z_vals=None, weights=None, isFine=False, *, t_rand=None, u=None):
    """Sample depths and build the network input (nerf_process.py:32-85).
    Returns ``(embedded [n*S, 90], z_vals [n, S], rays_d [n, 3])``."""
    rays = as_f32_dev(rays)
    n = rays.shape[0]
    L_x, L_d = int(getattr(posenc[0], "L", 10)), int(getattr(posenc[1], "L", 4))
    if not isFine:
        if t_rand is None:
            t_rand = ops.fill_uniform(_next_seed(None), 0, 0, n, int(opts.N_samples_c), rays.device)
        z = ops.stratified_z(float(opts.near), float(opts.far), as_f32_dev(t_rand, rays.device))
    else:
        det = _det(opts)
        if not det and u is None:
            u = ops.fill_uniform(_next_seed(None), 1, 0, n, int(opts.N_samples_f), rays.device)
        z = ops.fine_z(as_f32_dev(z_vals, rays.device), as_f32_dev(weights, rays.device), int(opts.N_samples_f), det,
                       None if det else as_f32_dev(u, rays.device))
    embedded = ops.embed(rays, z, L_x, L_d)
    return embedded, z, rays[:, 3:]


class _Composite(torch.autograd.Function):
    """post_process with the gradient the reference's loss uses: d rgb_map -> d raw (mi_nerf_composite_backward).
    disp / acc / weights / depth are returned without a graph (train.py:60-66 reads the colours only)."""

    @staticmethod
    def forward(ctx, raw, z, d):
        rgb, disp, acc, wts, depth = ops.composite(raw, z, d, want_all=True)
        ctx.save_for_backward(raw, z, d)
        ctx.mark_non_differentiable(disp, acc, wts, depth)
        ctx.set_materialize_grads(False)
        return rgb, disp, acc, wts, depth

    @staticmethod
    def backward(ctx, g_rgb, *_):
        if g_rgb is None:
            return None, None, None
        raw, z, d = ctx.saved_tensors
        return ops.composite_backward(raw, z, d, g_rgb.contiguous().float()), None, None


def post_process(outputs, z_vals, rays_d):
    """Alpha compositing (nerf_process.py:89-140) -> (rgb_map, disp_map, acc_map, weights, depth_map).
    Differentiable in ``rgb_map`` w.r.t. ``outputs`` when they carry a graph."""
    z = as_f32_dev(z_vals)
    raw = as_f32_dev(outputs, z.device)
    d = as_f32_dev(rays_d, z.device)
    if torch.is_grad_enabled() and raw.requires_grad:
        return _Composite.apply(raw, z.detach(), d.detach())
    return ops.composite(raw, z, d, want_all=True)


raw2outputs = post_process          # north-star alias (original NeRF naming)


def run_network(model, embedded, is_fine: bool = False):
    """north-star alias: the chunked ``model(embedded)`` loop of nerf_process.py:190-192,206-207 as one launch."""
    if train_path.wants_grad(model):
        return model(embedded, is_fine)                          # differentiable route of the model mirror (model/NeRF.py)
    packed = packed_for(model)
    return ops.mlp_embedded(packed.net, packed.blob(is_fine), as_f32_dev(embedded, packed.device))


# --------------------------------------------------------------------------------------------------
def _render(rays: torch.Tensor, packed: PackedNeRF, opts, t_rand, u, seed: int, ray_offset: int, bf16: bool,
            intermediates: bool, f16s: bool = False, coarse_f16s: bool = False) -> Dict[str, torch.Tensor]:
    n = rays.shape[0]
    dev = rays.device
    Sc, Nf = int(opts.N_samples_c), int(opts.N_samples_f)
    det = _det(opts)
    # jitter: explicit tensors when the caller injects them (or wants them back); otherwise the kernels draw it themselves from the
    # same counter-based generator, keyed on (seed, ray_offset + ray, sample) -- identical values, no tensors, no extra launches
    cfg = ops.render_cfg(float(opts.near), float(opts.far), Sc, Nf, det, bf16, seed=seed, ray_offset=ray_offset, f16s=f16s, coarse_f16s=coarse_f16s)
    if t_rand is not None:
        t_rand = as_f32_dev(t_rand, dev)
    elif intermediates:
        t_rand = ops.fill_uniform(seed, 0, ray_offset, n, Sc, dev)
    if Nf > 0 and not det:
        if u is not None:
            u = as_f32_dev(u, dev)
        elif intermediates:
            u = ops.fill_uniform(seed, 1, ray_offset, n, Nf, dev)
    else:
        u = None
    if coarse_f16s:                                                 # MI_NERF_MODE_F16S_BF16: one blob of each family (both 256 wide)
        blobs = (packed.f16s()[0], packed.bf16()[1])
    else:
        blobs = packed.f16s() if f16s else (packed.bf16() if bf16 else (packed.coarse, packed.fine))
    rgb_c, disp_c, rgb_f, disp_f, ws = ops.render_rays(packed.kernel_net(bf16, f16s), blobs[0], blobs[1] if Nf > 0 else None, cfg, rays, t_rand, u)
    out = {"rgb_c": rgb_c, "disp_c": disp_c}                        # nerf_process.py:215-216
    if Nf > 0:
        out["rgb_f"], out["disp_f"] = rgb_f, disp_f
    if intermediates:
        for k, v in ops.workspace_views(cfg, n, ws).items():
            out["_" + k] = v
        out["_t_rand"], out["_u"] = t_rand, u
    return out


def render_rays(rays, model, posenc, opts, *, t_rand=None, u=None, seed=None, ray_offset: int = 0, bf16: bool = False,
                return_intermediates: bool = False, f16s: bool = False, coarse_f16s: bool = False):
    """Coarse pass -> composite -> resample -> fine pass (nerf_process.py:185-216) as one fused launch
    sequence.  Returns ``{'rgb_c','disp_c'[,'rgb_f','disp_f']}``.  ``bf16`` / ``f16s`` select the network's precision mode
    (fp32 MFMA by default; bf16 MFMA; f16 split precision: fp32-grade results on the f16 matrix pipe); ``bf16`` with ``coarse_f16s``
    evaluates the coarse network in split precision and the fine one in bf16 (the fine sample positions then match the fp32 path's)."""
    if train_path.wants_grad(model):
        if bf16 or return_intermediates:
            raise MiNerfError("the training path has no bf16 mode (fp32, or f16s=True: split-precision forward) and returns no intermediates")
        if rays.dim() != 2 or rays.shape[1] != 6:
            raise MiNerfError(f"rays must be [n, 6] (o, d), got {tuple(rays.shape)}")
        return train_path.render_train(rays, model, opts, t_rand=t_rand, u=u, seed=_next_seed(seed), ray_offset=int(ray_offset), f16s=f16s)
    packed = packed_for(model)
    rays = as_f32_dev(rays, packed.device)
    if rays.dim() != 2 or rays.shape[1] != 6:
        raise MiNerfError(f"rays must be [n, 6] (o, d), got {tuple(rays.shape)}")
    return _render(rays, packed, opts, t_rand, u, _next_seed(seed), int(ray_offset), bf16, return_intermediates, f16s, coarse_f16s)


def batchify_rays_and_render_by_chunk(ray_o, ray_d, model, posenc, H, W, K, opts, *, t_rand=None, u=None, seed=None,
                                      ray_offset: int = 0, bf16: bool = False, f16s: bool = False, coarse_f16s: bool = False):
    """Drop-in entry point (nerf_process.py:220-252): flatten, optional NDC warp for llff, render.
    Returns ``(rgb_c [N,3], disp_c [N], rgb_f [N,3] | None, disp_f [N] | None)``.

    ``opts.chunk_rays`` bounded the reference's activation memory; the fused kernels keep activations in
    registers, so rays are launched in slabs of up to MAX_RAYS_PER_LAUNCH.  The result does not depend on
    the slab size because the jitter is keyed on the global ray index (``ray_offset`` + position)."""
    training = train_path.wants_grad(model)
    if training and bf16:
        raise MiNerfError("the training path has no bf16 mode (fp32, or f16s=True: split-precision forward, fp32 backward)")
    packed = None if training else packed_for(model)
    dev = next(model.parameters()).device if training else packed.device
    ray_d = as_f32_dev(ray_d, dev)
    flat_d = ray_d.reshape(-1, 3)
    ray_o = ray_o.to(dev) if ray_o.device != dev else ray_o
    flat_o = ray_o.to(torch.float32).expand(ray_d.shape).reshape(-1, 3)     # nerf_process.py:221 (accepts the stride-0 view)
    if getattr(opts, "data_type", None) == "llff":                  # nerf_process.py:224-226
        k00 = K[0][0]
        focal = float(k00.item()) if isinstance(k00, torch.Tensor) else float(k00)
        flat_o, flat_d = ndc_rays(H, W, focal, 1.0, flat_o, flat_d)
    rays = torch.cat((flat_o, flat_d), dim=-1)                      # nerf_process.py:229
    N = rays.shape[0]
    seed = _next_seed(seed)
    Nf = int(opts.N_samples_f)
    parts = []
    slab = train_path.MAX_TRAIN_RAYS if training else MAX_RAYS_PER_LAUNCH
    for i in range(0, N, slab):
        j = min(N, i + slab)
        tr, uu = (None if t_rand is None else t_rand[i:j]), (None if u is None else u[i:j])
        if training:                                                # train.py:53-54: one autograd node per slab
            parts.append(train_path.render_train(rays[i:j].contiguous(), model, opts, t_rand=tr, u=uu, seed=seed,
                                                 ray_offset=int(ray_offset) + i, f16s=f16s))
        else:
            parts.append(_render(rays[i:j], packed, opts, tr, uu, seed, int(ray_offset) + i, bf16, False, f16s, coarse_f16s))
    def cat(key):
        return parts[0][key] if len(parts) == 1 else torch.cat([p[key] for p in parts], dim=0)
    if Nf > 0:
        return cat("rgb_c"), cat("disp_c"), cat("rgb_f"), cat("disp_f")
    return cat("rgb_c"), cat("disp_c"), None, None
