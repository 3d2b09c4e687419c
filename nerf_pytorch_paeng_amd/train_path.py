"""Training step of the hot path -- what the reference gets from autograd for train.py:53-70.

``batchify_rays_and_render_by_chunk`` routes here when gradients are enabled and the model's parameters
require them.  The forward runs the same kernels as inference (the MLP in its STASH variant, which also
keeps every layer's activations), wrapped in ONE ``torch.autograd.Function`` whose inputs are the model's
parameters; the backward is hand-written HIP (composite backward, backward-data chain, backward-weights)
and returns one gradient per parameter, so ``loss.backward(); optimizer.step()`` work unchanged
(train.py:69-70).  PyTorch does the bookkeeping (graph node, parameter ``.grad`` accumulation, the
optimizer); no PyTorch kernel touches the activations.

Semantics copied from the reference: only ``rgb_c`` / ``rgb_f`` carry gradient; the fine depths are
detached (nerf_process.py:66), so the fine loss does not reach the coarse network; disparity outputs are
not differentiable here (the reference's loss never reads them, train.py:60-66).
"""
from __future__ import annotations

import weakref
from typing import Dict, List, Optional

import torch

from . import ops
from ._lib import MiNerfError, Net, as_f32_dev
from .weights import _param_shape, infer_net, pad_index_map, padded_train_net

# rays per autograd node: the activation stash is ~9.9 KB per sample point (W = 256, D = 8)
MAX_TRAIN_RAYS = 16384

# The split-precision backward runs on d_raw * s with ~256x of f16 headroom (FP16_OVFL: conversions clip at 65504, never inf), and the
# split-precision packers cannot represent a weight beyond the f16 range: both failure modes are silent on the device.  Every f16s backward
# folds what it saw into three device words of the model's state (no host synchronisation); they are READ -- one device -> host copy -- on
# after the LAST backward launch of the first F16S_CHECK_FIRST training steps (a step = one _RenderTrain.backward: the coarse and the fine net's
# launches together, so the fine net's words are in the first read) and then every F16S_CHECK_EVERY-th step, and a saturated chain or an
# unrepresentable weight then raises (or warns, by F16S_ON_SATURATION).  By the time a later read raises, up to F16S_CHECK_EVERY - 1
# optimizer.step() calls have already applied clipped gradients: the message says so.  f16s_status(model) reads the words on demand (without
# resetting them unless asked); harness.train() returns them in its dict.
F16S_CHECK_EVERY = 50
F16S_CHECK_FIRST = 3
F16S_ON_SATURATION = "raise"          # or "warn"
F16_MAX = 65504.0


class _TrainState:
    """Per-model constants of the training path: network shape and the device-side pack maps."""

    def __init__(self, model: torch.nn.Module, device: torch.device, f16s: bool = False):
        sd = model.state_dict()
        # net_model: the module's own shape (model/NeRF.py:24-30); net: what the training kernels run -- the same, or for a width without
        # training kernels (--netWidth 64, config.py:57) the next wider one, parameters scattered into zeros (weights.pad_index_map)
        self.net_model: Net = infer_net(sd)
        # ... which depends on the precision: the split-precision kernels exist for W = 256 only (one state per kernel width, _state_for)
        self.net: Net = padded_train_net(self.net_model, f16s)
        self.pad_idx = None if self.net is self.net_model else pad_index_map(self.net_model, self.net).to(device)
        self.n_flat = ops.param_count(self.net)
        self.device = device
        self.names = ops.param_names(self.net)
        self.map_fwd = ops.pack_map(self.net, False).to(device)
        self.map_bwd = ops.pack_map(self.net, True).to(device)
        self._map_f16s = None
        self.f16s_out_of_range = None          # device counter: weights the split-precision packer could not represent (NaN / beyond f16)
        self.f16s_range = None                 # device [2]: running max of (max |d_raw|, max |delta * s|) over the f16s backwards since the last read
        self.f16s_steps = 0                    # f16s training steps (backward passes) so far (host count; sets the read cadence)
        self.f16s_steps_since_read = 0

    def map_f16s(self) -> torch.Tensor:
        """Gather map of the split-precision blob, built on first use (f16s=True training forward)."""
        if self._map_f16s is None:
            self._map_f16s = ops.pack_map_f16s(self.net).to(self.device)
            self._map_bwd_f16s = ops.pack_map_f16s(self.net, backward=True).to(self.device)
            self.f16s_out_of_range = torch.zeros(1, dtype=torch.int32, device=self.device)
            self.f16s_range = torch.zeros(2, dtype=torch.float32, device=self.device)
        return self._map_f16s

    def note_f16s_backward(self, work: torch.Tensor, n_rays: int, S: int) -> None:
        """Fold the range words the split-precision backward left in ``work`` into the running maximum (device side; no host sync)."""
        if n_rays == 0:
            return                                 # an empty slab launches nothing: there are no words to fold
        torch.maximum(self.f16s_range, ops.backward_range_words(self.net, n_rays, S, work), out=self.f16s_range)

    def end_f16s_step(self) -> None:
        """After the last backward launch of a training step: at the cadence, read the words and complain."""
        self.f16s_steps += 1
        self.f16s_steps_since_read += 1
        if self.f16s_steps <= int(F16S_CHECK_FIRST) or self.f16s_steps % max(1, int(F16S_CHECK_EVERY)) == 0:
            self.check_f16s()

    def read_f16s(self, reset: bool = True) -> Dict[str, float]:
        """One device -> host read of the three words: {"max_abs_d_raw", "max_abs_delta_scaled", "weights_out_of_range", "saturated"}."""
        if self.f16s_range is None:
            return {"max_abs_d_raw": 0.0, "max_abs_delta_scaled": 0.0, "weights_out_of_range": 0, "saturated": False}
        both = torch.cat([self.f16s_range, self.f16s_out_of_range.float()]).cpu()
        if reset:
            self.f16s_range.zero_()
            self.f16s_out_of_range.zero_()
            self.f16s_steps_since_read = 0
        d, ds, oor = float(both[0]), float(both[1]), int(both[2])
        return {"max_abs_d_raw": d, "max_abs_delta_scaled": ds, "weights_out_of_range": oor, "saturated": bool(ds >= F16_MAX or ds != ds or oor > 0)}

    def check_f16s(self) -> Dict[str, float]:
        steps = self.f16s_steps_since_read
        r = self.read_f16s()
        if r["saturated"]:
            msg = (f"split-precision training step out of range: max |delta * s| = {r['max_abs_delta_scaled']:.4g} (f16 max {F16_MAX:.0f}; max |d_raw| = "
                   f"{r['max_abs_d_raw']:.4g}), {r['weights_out_of_range']} weight(s) beyond the f16 range, within the last {steps} training step(s) "
                   f"(this one included): their gradients were clipped, and the optimizer has already applied up to {max(0, steps - 1)} of them.  "
                   f"Train with the fp32 backward (precision 'fp32') or scale the loss down.")
            if F16S_ON_SATURATION == "warn":
                import warnings
                warnings.warn(msg, RuntimeWarning, stacklevel=3)
            else:
                raise MiNerfError(msg)
        return r

    def map_bwd_f16s(self) -> torch.Tensor:
        self.map_f16s()
        return self._map_bwd_f16s

    def flat(self, params) -> torch.Tensor:
        """Flat fp32 parameter vector of the network the kernels run (zero-padded to its width when the module's is narrower)."""
        f = _flat(params)
        if self.pad_idx is None:
            return f
        wide = torch.zeros(self.n_flat, dtype=torch.float32, device=f.device)
        wide[self.pad_idx] = f
        return wide

    def split_grads(self, grads: torch.Tensor) -> List[torch.Tensor]:
        """The kernels' flat gradient -> one tensor per parameter of the MODULE, in st.names order."""
        if self.pad_idx is not None:
            grads = grads[self.pad_idx]
        out, off = [], 0
        for k in self.names:
            shape = _param_shape(self.net_model, k)
            cnt = 1
            for s in shape:
                cnt *= s
            out.append(grads[off:off + cnt].view(shape))
            off += cnt
        return out

    def params(self, module: torch.nn.Module) -> List[torch.Tensor]:
        named = dict(module.named_parameters())
        try:
            return [named[k] for k in self.names]
        except KeyError as e:
            raise MiNerfError(f"module lacks parameter {e} (expected the layout of model/NeRF.py:24-30)") from e


# model -> {kernel width: state}: a netWidth below 128 trains 128 wide in fp32 and 256 wide in split precision
_states: "weakref.WeakKeyDictionary[torch.nn.Module, Dict[int, _TrainState]]" = weakref.WeakKeyDictionary()


def _state_for(model: torch.nn.Module, f16s: bool = False) -> _TrainState:
    dev = next(model.parameters()).device
    if dev.type != "cuda":
        raise MiNerfError(f"model lives on {dev}: the MI355X path needs a HIP device (no CPU fallback)")
    per_width = _states.get(model)
    if per_width is None:
        per_width = _states[model] = {}
    known = next(iter(per_width.values()), None)
    if known is not None:                                  # the module's own shape is on record: which kernel width does this precision train at?
        st = per_width.get(padded_train_net(known.net_model, f16s).W)
        if st is not None and st.device == dev:
            return st
    st = _TrainState(model, dev, f16s)
    per_width[st.net.W] = st
    return st


def wants_grad(model) -> bool:
    return (torch.is_grad_enabled() and isinstance(model, torch.nn.Module)
            and any(p.requires_grad for p in model.parameters()))


def _flat(params) -> torch.Tensor:
    return torch.cat([p.detach().reshape(-1) for p in params]).float()


class _RenderTrain(torch.autograd.Function):
    """rays (+ explicit randomness) and the two networks' parameters -> rgb_c, disp_c, rgb_f, disp_f."""

    @staticmethod
    def forward(ctx, st: _TrainState, rays, cfg: Dict, t_rand, u, z_override, *params):
        n_each = len(st.names)
        net = st.net
        Sc, Nf, det = cfg["Sc"], cfg["Nf"], cfg["det"]
        f16s = bool(cfg.get("f16s", False))

        def forward_net(flat, blob, z):
            # f16s: the forward launch runs in split precision (fp32-grade outputs and stash; blob re-packed on the device like the fp32
            # one); the backward is the same fp32 kernels either way and keeps reading the fp32 blob's side tables
            if f16s:
                return ops.mlp_rays_train(net, ops.pack_apply_f16s(net, st.map_f16s(), flat, st.f16s_out_of_range), rays, z, f16s=True)
            return ops.mlp_rays_train(net, blob, rays, z)

        flat_c = st.flat(params[:n_each])
        blob_c = ops.pack_apply(st.map_fwd, flat_c)
        z_c = ops.stratified_z(cfg["near"], cfg["far"], t_rand) if z_override is None else z_override[0]
        raw_c, stash_c = forward_net(flat_c, blob_c, z_c)
        rgb_c, disp_c, _, w_c, _ = ops.composite(raw_c, z_c, rays, want_all=True)
        ctx.st, ctx.Nf, ctx.f16s = st, Nf, f16s
        saved = [rays, flat_c, blob_c, z_c, raw_c, stash_c]
        if Nf > 0:
            flat_f = st.flat(params[n_each:])
            blob_f = ops.pack_apply(st.map_fwd, flat_f)
            z_f = ops.fine_z(z_c, w_c, Nf, det, None if det else u) if (z_override is None or z_override[1] is None) else z_override[1]
            raw_f, stash_f = forward_net(flat_f, blob_f, z_f)
            rgb_f, disp_f, *_ = ops.composite(raw_f, z_f, rays, want_all=False)
            saved += [flat_f, blob_f, z_f, raw_f, stash_f]
        else:
            rgb_f = torch.empty(0, 3, device=rays.device)
            disp_f = torch.empty(0, device=rays.device)
        ctx.save_for_backward(*saved)
        ctx.mark_non_differentiable(disp_c, disp_f)
        ctx.set_materialize_grads(False)
        return rgb_c, disp_c, rgb_f, disp_f

    @staticmethod
    def backward(ctx, g_rgb_c, g_disp_c, g_rgb_f, g_disp_f):
        st: _TrainState = ctx.st
        net = st.net
        saved = ctx.saved_tensors
        rays = saved[0]

        def one(flat, blob, z, raw, stash, g_rgb) -> List[Optional[torch.Tensor]]:
            if g_rgb is None:
                return [None] * len(st.names)
            f16s = ctx.f16s and net.W == 256
            f16s_dgrad = f16s and net.D <= 15              # the split-precision chain keeps a tile's ReLU' words of all layers in LDS
            blob_b = ops.pack_apply_f16s(net, st.map_bwd_f16s(), flat, st.f16s_out_of_range, backward=True) if f16s_dgrad else ops.pack_apply(st.map_bwd, flat)
            d_raw = ops.composite_backward(raw, z, rays, g_rgb.contiguous().float())
            grads, work = ops.mlp_backward(net, blob, blob_b, rays, z, d_raw, stash, f16s_wgrad=f16s, f16s_dgrad=f16s_dgrad)
            if f16s:
                st.note_f16s_backward(work, z.shape[0], z.shape[1])
            return st.split_grads(grads)

        gc = one(*saved[1:6], g_rgb_c)
        gf = one(*saved[6:11], g_rgb_f) if ctx.Nf > 0 else [None] * len(st.names)
        if ctx.f16s and net.W == 256 and (g_rgb_c is not None or g_rgb_f is not None):
            st.end_f16s_step()                                 # both nets' range words are folded: read them at the cadence
        return (None, None, None, None, None, None, *gc, *gf)


class _EmbeddedTrain(torch.autograd.Function):
    """model(embedded, is_fine) with gradients: pre-embedded rows [n, 90] and ONE network's parameters -> [n, 4]."""

    @staticmethod
    def forward(ctx, st: _TrainState, x, *params):
        flat = st.flat(params)
        blob = ops.pack_apply(st.map_fwd, flat)
        out, stash = ops.mlp_embedded_train(st.net, blob, x)
        ctx.st = st
        ctx.save_for_backward(x, flat, blob, stash)
        return out

    @staticmethod
    def backward(ctx, g_out):
        st: _TrainState = ctx.st
        x, flat, blob, stash = ctx.saved_tensors
        blob_b = ops.pack_apply(st.map_bwd, flat)
        grads = ops.mlp_embedded_backward(st.net, blob, blob_b, x, g_out.contiguous().float(), stash)
        return (None, None, *st.split_grads(grads))


def render_train(rays: torch.Tensor, model: torch.nn.Module, opts, *, t_rand=None, u=None, seed: int = 0, ray_offset: int = 0,
                 z_override=None, det: Optional[bool] = None, f16s: bool = False) -> Dict[str, torch.Tensor]:
    """Differentiable ``render_rays`` (nerf_process.py:185-216) for one slab of rays [n, 6].  ``f16s``: the two forward launches run in
    split precision (fp32-grade results, ~3x faster); the backward kernels are the fp32 ones."""
    st = _state_for(model, f16s)                 # f16s: the 256-wide state (any netWidth <= 256 is scattered into it)
    dev = st.device
    if isinstance(rays, torch.Tensor) and rays.requires_grad:
        raise MiNerfError("rays require grad: the training path differentiates w.r.t. the MLP parameters only (the reference trains "
                          "nothing else, main.py:79-80); detach the rays, or a gradient would be dropped silently")
    rays = as_f32_dev(rays, dev)
    n = rays.shape[0]
    Sc, Nf = int(opts.N_samples_c), int(opts.N_samples_f)
    if det is None:
        p = getattr(opts, "perturb", 1.0)
        det = isinstance(p, (int, float)) and p == 0.0
    t_rand = ops.fill_uniform(seed, 0, ray_offset, n, Sc, dev) if t_rand is None else as_f32_dev(t_rand, dev)
    if Nf > 0 and not det:
        u = ops.fill_uniform(seed, 1, ray_offset, n, Nf, dev) if u is None else as_f32_dev(u, dev)
    else:
        u = None
    cfg = {"near": float(opts.near), "far": float(opts.far), "Sc": Sc, "Nf": Nf, "det": bool(det), "f16s": bool(f16s)}
    params = st.params(model.model_coarse) + st.params(model.model_fine)
    rgb_c, disp_c, rgb_f, disp_f = _RenderTrain.apply(st, rays, cfg, t_rand, u, z_override, *params)
    out = {"rgb_c": rgb_c, "disp_c": disp_c}
    if Nf > 0:
        out["rgb_f"], out["disp_f"] = rgb_f, disp_f
    return out


def f16s_status(model: torch.nn.Module, reset: bool = True) -> Dict[str, float]:
    """What the split-precision training steps of ``model`` saw since the last read (one device -> host copy): max |d_raw|, max |delta * s|
    (how much of the f16 range the scaled backward chain used; >= 65504 means a conversion saturated and gradients were clipped), the number
    of weights the split-precision packers could not represent, and ``saturated``.  The training path itself reads these after each of the first
    F16S_CHECK_FIRST f16s steps and every F16S_CHECK_EVERY-th after them, and raises / warns; this is the on-demand read."""
    st = next((s for s in (_states.get(model) or {}).values() if s.f16s_range is not None), None)
    if st is None:
        return {"max_abs_d_raw": 0.0, "max_abs_delta_scaled": 0.0, "weights_out_of_range": 0, "saturated": False}
    return st.read_f16s(reset)


def rank_seed(seed: int, rank: int) -> int:
    """Per-rank jitter seed: ``seed`` mixed with a hash of the rank (distinct for every rank below 2^32, rank 0 included)."""
    h = ((rank + 1) * 0x9E3779B1) & 0xFFFFFFFF
    h ^= h >> 15
    h = (h * 0x85EBCA77) & 0xFFFFFFFF
    h ^= h >> 13
    return (seed ^ h) & 0xFFFFFFFF


class RenderModule(torch.nn.Module):
    """``forward(rays_o, rays_d, H, W, K)`` = ``batchify_rays_and_render_by_chunk`` on the wrapped model.

    The reference trains on one GPU (main.py:166-170).  For data-parallel training wrap THIS module, not the NeRF, in
    ``torch.nn.parallel.DistributedDataParallel``: DDP needs the forward to go through the wrapper, and the NeRF's own
    ``forward(x, is_fine)`` is the embedded-input call of model/NeRF.py:70-78.  The hand-written backward returns ordinary
    parameter gradients and DDP all-reduces them in buckets (RCCL over xGMI with backend "nccl").  Both networks' gradients
    come out of ONE autograd node, so they become ready together at the end of the backward: the all-reduce of the 4.8 MB of
    gradients (~0.1 ms on xGMI) follows the 20 ms backward, it does not overlap with it.

    Jitter: without explicit ``t_rand`` / ``u`` / ``ray_offset`` / ``seed`` the generator is keyed on (seed, local ray index);
    every rank would then draw the SAME jitter for its local ray i.  ``forward`` therefore folds the rank into the SEED
    (not into the 32-bit ray counter, which would wrap for large rank x batch products) whenever one of the two random
    tensors is generated internally and the caller pins neither ``ray_offset`` nor ``seed``."""

    def __init__(self, model: torch.nn.Module, posenc, opts):
        super().__init__()
        self.model, self.posenc, self.opts = model, posenc, opts

    def forward(self, rays_o, rays_d, H, W, K, **kw):
        from . import nerf_process as NP
        generates = kw.get("t_rand") is None or (kw.get("u") is None and int(self.opts.N_samples_f) > 0)
        if generates and "ray_offset" not in kw and "seed" not in kw:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                kw["seed"] = rank_seed(NP._next_seed(None), dist.get_rank())
        return NP.batchify_rays_and_render_by_chunk(rays_o, rays_d, self.model, self.posenc, H, W, K, self.opts, **kw)
