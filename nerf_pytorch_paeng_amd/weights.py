"""Weight ingest: reference checkpoint layout -> device-resident packed blobs.

The reference keeps its parameters in an ``nn.Module`` whose ``state_dict`` has the keys
``model_{coarse,fine}.linear_x.{i}.{weight,bias}``, ``linear_d``, ``linear_feat``, ``linear_density``,
``linear_color`` (model/NeRF.py:24-30,58-59; saved at train.py:105-114, loaded at test.py:20-21).
``PackedNeRF`` accepts such a state dict (numpy or torch), infers (D, W, skip, L_x, L_d) from the
shapes, packs both networks into the kernels' streaming layout and keeps the blobs on the device.
"""
from __future__ import annotations

import weakref
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import ops
from ._lib import MiNerfError, Net


def infer_net(sd: Dict[str, "np.ndarray | torch.Tensor"], prefix: str = "model_coarse.") -> Net:
    """(D, W, skip, L_x, L_d) from parameter shapes (the construction rule of model/NeRF.py:24-30)."""
    def shape(k):
        return tuple(sd[prefix + k].shape)
    D = 0
    while f"{prefix}linear_x.{D}.weight" in sd:
        D += 1
    if D == 0:
        raise MiNerfError(f"no {prefix}linear_x.0.weight in state dict")
    W, in_x = shape("linear_x.0.weight")
    skip = -1
    for l in range(1, D):
        fan_in = shape(f"linear_x.{l}.weight")[1]
        if fan_in == W + in_x:
            if skip != -1:
                raise MiNerfError("more than one skip connection is not supported")
            skip = l - 1
        elif fan_in != W:
            raise MiNerfError(f"unexpected fan-in {fan_in} at trunk layer {l}")
    in_d = shape("linear_d.weight")[1] - W
    if (in_x - 3) % 6 or (in_d - 3) % 6:
        raise MiNerfError(f"input widths {in_x}/{in_d} are not 3+6L")
    return ops.make_net(D, W, skip, (in_x - 3) // 6, (in_d - 3) // 6)


def _param_shape(net: Net, key: str) -> Tuple[int, ...]:
    """Shape of one parameter of a NeRFModule (model/NeRF.py:24-30) from the network description."""
    W, in_x, in_d = net.W, 3 + 6 * net.L_x, 3 + 6 * net.L_d
    mod, kind = key.rsplit(".", 1)
    if mod.startswith("linear_x."):
        l = int(mod.split(".")[1])
        fan_in = in_x if l == 0 else (W + in_x if (net.skip >= 0 and l == net.skip + 1) else W)
        out = W
    else:
        out, fan_in = {"linear_d": (W // 2, W + in_d), "linear_feat": (W, W), "linear_density": (1, W), "linear_color": (3, W // 2)}[mod]
    return (out, fan_in) if kind == "weight" else (out,)


def padded_train_net(net: Net, f16s: bool = False) -> Net:
    """The network the TRAINING kernels run for ``net``: the fp32 kernels exist for W = 128 and 256, the split-precision ones (``f16s``) for
    W = 256 only; a narrower network trains as the next width that has kernels -- so the width depends on the precision: netWidth 64 trains
    128 wide in fp32 and 256 wide in split precision -- with zero weights and biases for the hidden units it does not have (``pad_index_map``)."""
    if net.W == 256 or (net.W == 128 and not f16s):
        return net
    if not 2 <= net.W < 256:
        raise MiNerfError(f"the training kernels exist for netWidth <= 256 (got {net.W}); wider networks run inference only")
    return ops.make_net(net.D, 256 if (f16s or net.W > 128) else 128, net.skip, net.L_x, net.L_d)


def padded_256_net(net: Net, what: str) -> Net:
    """The network the bf16 / split-precision kernels run for ``net``: they exist for W = 256; a narrower network runs as a 256-wide one with
    zero weights for the hidden units it does not have (``pad_index_map``; the fp32 kernels get the same from the C packer)."""
    if net.W == 256:
        return net
    if not 2 <= net.W < 256:
        raise MiNerfError(f"the {what} variant is built for netWidth <= 256 (got {net.W}); wider networks run in fp32")
    return ops.make_net(net.D, 256, net.skip, net.L_x, net.L_d)


def padded_state_dict(sd, prefix: str, net: Net, wide: Net) -> Dict[str, np.ndarray]:
    """One module's parameters (``prefix`` + linear_x.0.weight ...) scattered into the shapes of ``wide`` (zeros elsewhere), keys kept."""
    flat = np.concatenate([np.asarray(sd[prefix + k].detach().cpu() if isinstance(sd[prefix + k], torch.Tensor) else sd[prefix + k],
                                      dtype=np.float32).reshape(-1) for k in ops.param_names(net)])
    out_flat = np.zeros(ops.param_count(wide), dtype=np.float32)
    out_flat[pad_index_map(net, wide).numpy()] = flat
    out, off = {}, 0
    for k in ops.param_names(wide):
        shp = _param_shape(wide, k)
        cnt = int(np.prod(shp))
        out[prefix + k] = out_flat[off:off + cnt].reshape(shp)
        off += cnt
    return out


def pad_index_map(net: Net, wide: Net) -> torch.Tensor:
    """int64 [param_count(net)]: where each entry of ``net``'s flat parameter vector (module.parameters() order) sits in the flat vector of
    ``wide`` = the same network at a larger width.  Rows keep their index; so do columns, except linear_d's view-direction block, which
    follows the (wider) feature block (model/NeRF.py:28,46: cat([feature, gamma(d)])).  Scattering a network's parameters through this map
    into zeros gives a network that computes the same function: the extra units are exactly 0 through their ReLU.  Their gradients are 0
    too (ReLU'(0) = 0 on the way in, activation 0 on the way out), so gathering the wide gradient through the same map IS the gradient."""
    ck = tuple(getattr(n, f) for n in (net, wide) for f, _ in Net._fields_)
    if ck in _pad_maps:
        return _pad_maps[ck]
    idx, off = [], 0
    for key in ops.param_names(net):
        shp, shp_w = _param_shape(net, key), _param_shape(wide, key)
        if len(shp) == 1:
            idx.append(off + torch.arange(shp[0]))
        else:
            cols = torch.arange(shp[1])
            if key == "linear_d.weight":
                cols = torch.where(cols < net.W, cols, cols + (wide.W - net.W))
            idx.append((off + torch.arange(shp[0])[:, None] * shp_w[1] + cols[None, :]).reshape(-1))
        n = 1
        for v in shp_w:
            n *= v
        off += n
    _pad_maps[ck] = torch.cat(idx).to(torch.int64)
    return _pad_maps[ck]


_pad_maps: Dict[tuple, torch.Tensor] = {}          # (net, wide) -> CPU index map (a few MB at most; built once per pair of shapes)


class PackedNeRF:
    """Both networks of a NeRF, packed and resident on one device."""

    def __init__(self, net: Net, coarse: torch.Tensor, fine: torch.Tensor):
        self.net, self.coarse, self.fine = net, coarse, fine
        self._bf16: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
        self._f16s: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
        self._sd = None
        self._flat: Optional[Tuple[torch.Tensor, torch.Tensor]] = None      # device-resident flat parameter vectors (nn.Module source)
        self.f16s_out_of_range: Optional[torch.Tensor] = None               # device int32 [1], set by the device-side f16s packer
        self._range_token = None     # (module, parameter-version key) when packed from an nn.Module: lets f16s() reuse a range verdict (_f16s_verdicts)

    @property
    def device(self) -> torch.device:
        return self.coarse.device

    @classmethod
    def from_state_dict(cls, sd, device, keep_state: bool = True) -> "PackedNeRF":
        net = infer_net(sd)
        fine_net = infer_net(sd, "model_fine.")
        if tuple(getattr(net, f) for f, _ in Net._fields_) != tuple(getattr(fine_net, f) for f, _ in Net._fields_):
            raise MiNerfError("coarse and fine networks differ in shape")
        c = ops.pack_module(sd, "model_coarse.", net).to(device)
        f = ops.pack_module(sd, "model_fine.", net).to(device)
        self = cls(net, c, f)
        if keep_state:
            self._sd = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in sd.items()
                        if k.startswith("model_")}
        return self

    def kernel_net(self, bf16: bool = False, f16s: bool = False) -> Net:
        """The network description to hand to the library WITH the blobs of that mode: the network's own for fp32 (the C packer pads it to
        a kernel width itself), the 256-wide one for the bf16 / split-precision blobs of a narrower network (padded here)."""
        if bf16 or f16s:
            return padded_256_net(self.net, "split-precision" if f16s else "bf16")
        return self.net

    def _wide_flats(self, wide: Net):
        """The device-resident flat parameter vectors scattered into the layout of ``wide`` (nn.Module source)."""
        if wide is self.net:
            return self._flat
        idx = pad_index_map(self.net, wide).to(self.device)
        out = []
        for flat in self._flat:
            w = torch.zeros(ops.param_count(wide), dtype=torch.float32, device=self.device)
            w[idx] = flat
            out.append(w)
        return tuple(out)

    def _wide_sd(self, wide: Net):
        if wide is self.net:
            return self._sd
        return {**padded_state_dict(self._sd, "model_coarse.", self.net, wide), **padded_state_dict(self._sd, "model_fine.", self.net, wide)}

    def bf16(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """bf16-stream blobs for the bf16 MFMA variant, packed lazily: on the device from the flat parameter vectors when this
        PackedNeRF came from an nn.Module (packed_for() makes a new one per call, so a host round trip here would be paid per
        call), on the host from the kept state dict otherwise (once).  Pass them to the library with ``kernel_net(bf16=True)``."""
        if self._bf16 is None:
            wide = self.kernel_net(bf16=True)
            if self._flat is not None:
                key = (tuple(getattr(wide, f) for f, _ in Net._fields_), str(self.device))
                if key not in _maps_bf16:
                    _maps_bf16[key] = ops.pack_map_bf16(wide).to(self.device)
                self._bf16 = tuple(ops.pack_apply_bf16(wide, _maps_bf16[key], flat) for flat in self._wide_flats(wide))
            else:
                if self._sd is None:
                    raise MiNerfError("bf16 packing needs the state dict (keep_state=True)")
                sd = self._wide_sd(wide)
                self._bf16 = (ops.pack_module(sd, "model_coarse.", wide, bf16=True).to(self.device),
                              ops.pack_module(sd, "model_fine.", wide, bf16=True).to(self.device))
        return self._bf16

    def f16s(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """Blobs of the split-precision variant (weights as f16 hi + lo pairs), packed lazily: on the device from the flat parameter
        vectors when this PackedNeRF came from an nn.Module (like bf16(): packed_for() makes a new one per call, so a host round trip
        here would be a synchronisation per render call), on the host from the kept state dict otherwise (once; the host packer refuses
        weights beyond the f16 range outright).  The device packer cannot refuse: it counts such weights into ``f16s_out_of_range``
        (a device int32), which ``check_f16s_range()`` reads -- here, right after the pack launches, ONCE PER VERSION of the module's
        parameters (a 4-byte device -> host read; later render calls on unchanged parameters reuse the verdict and do not synchronise;
        a frozen PackedNeRF pays it once): every caller of the split-precision mode gets the refusal the host packer gives."""
        if self._f16s is None:
            wide = self.kernel_net(f16s=True)
            if self._flat is not None:
                key = (tuple(getattr(wide, f) for f, _ in Net._fields_), str(self.device))
                if key not in _maps_f16s:
                    _maps_f16s[key] = ops.pack_map_f16s(wide).to(self.device)
                self.f16s_out_of_range = torch.zeros(1, dtype=torch.int32, device=self.device)
                blobs = tuple(ops.pack_apply_f16s(wide, _maps_f16s[key], flat, self.f16s_out_of_range) for flat in self._wide_flats(wide))
                # the 4-byte read synchronises the host: pay it once per parameter VERSION of the module, not once per render call
                # (packed_for() makes a new PackedNeRF per call).  A write through p.data does not bump the version; a weight pushed
                # out of range that way still cannot give a finite wrong colour -- it packs to NaN and the frame is NaN.
                module, version_key = self._range_token if self._range_token is not None else (None, None)
                if module is None or _f16s_verdicts.get(module) != version_key:
                    self.check_f16s_range()
                    if module is not None:
                        _f16s_verdicts[module] = version_key
                self._f16s = blobs
            else:
                if self._sd is None:
                    raise MiNerfError("f16-split packing needs the state dict (keep_state=True)")
                sd = self._wide_sd(wide)
                self._f16s = (ops.pack_module(sd, "model_coarse.", wide, f16s=True).to(self.device),
                              ops.pack_module(sd, "model_fine.", wide, f16s=True).to(self.device))
        return self._f16s

    def check_f16s_range(self) -> int:
        """Weights the device-side split-precision packer could not represent (NaN or beyond the f16 range) -- one device -> host read;
        raises when there are any.  f16s() calls it when it packs on the device; a caller may repeat it."""
        n = 0 if self.f16s_out_of_range is None else int(self.f16s_out_of_range.item())
        if n:
            raise MiNerfError(f"{n} weight(s) are NaN or beyond the f16 range (65504): the split-precision variant cannot carry this network; use precision 'fp32'")
        return n

    def blob(self, is_fine: bool) -> torch.Tensor:
        return self.fine if is_fine else self.coarse


# ---------------------------------------------------------------------------------------------------
# nn.Module models (ours or the reference's): re-packed from the LIVE parameters on every call
# ---------------------------------------------------------------------------------------------------
# An earlier version cached the packed blobs keyed on (data_ptr, _version, device) of every parameter.  Writes through ``p.data``
# (EMA updates, hand-written optimisers, some weight loaders) change the values without bumping ``_version`` or moving the storage,
# so the cache could serve stale weights to the no-grad path while the training path (which always re-packs) used the new ones.
# The device-side re-pack is one ``torch.cat`` of the 48 parameter tensors and two gather launches per network (tens of
# microseconds against the 8 ms of a 4096-ray batch, once per ``render_rays`` / ``batchify`` call), so it simply runs every time.
# Callers that render many batches from frozen weights pass a ``PackedNeRF`` (``PackedNeRF.from_state_dict`` or ``packed_for`` once).
def packed_for(model, device=None) -> PackedNeRF:
    """PackedNeRF for ``model``: a PackedNeRF passes through; an nn.Module with the reference's ``model_coarse`` /
    ``model_fine`` layout is packed from its current parameter values (never cached)."""
    if isinstance(model, PackedNeRF):
        return model
    if not isinstance(model, torch.nn.Module) or not hasattr(model, "model_coarse") or not hasattr(model, "model_fine"):
        raise MiNerfError("model must be a PackedNeRF or an nn.Module with model_coarse / model_fine (model/NeRF.py:58-59)")
    if device is None:
        device = next(model.parameters()).device
    if torch.device(device).type != "cuda":
        raise MiNerfError(f"model lives on {device}: the MI355X path needs a HIP device (no CPU fallback)")
    return _pack_module_on_device(model, torch.device(device))


# module -> the parameter-version key ((data_ptr, _version) of every parameter) whose split-precision range check came back clean
_f16s_verdicts: "weakref.WeakKeyDictionary[torch.nn.Module, tuple]" = weakref.WeakKeyDictionary()
# gather maps per network shape (built once by the host packer, kept on the device)
_maps: Dict[tuple, torch.Tensor] = {}
_maps_bf16: Dict[tuple, torch.Tensor] = {}
_maps_f16s: Dict[tuple, torch.Tensor] = {}


def _pack_module_on_device(model: torch.nn.Module, device: torch.device) -> PackedNeRF:
    """Re-pack after a parameter change without leaving the device (an optimizer.step() between two evaluations,
    main.py:140-149): flatten the parameters, gather them into the blob layout with the cached map."""
    sd = model.state_dict()
    net = infer_net(sd)
    fine_net = infer_net(sd, "model_fine.")
    if tuple(getattr(net, f) for f, _ in Net._fields_) != tuple(getattr(fine_net, f) for f, _ in Net._fields_):
        raise MiNerfError("coarse and fine networks differ in shape")
    key = (tuple(getattr(net, f) for f, _ in Net._fields_), str(device))
    if key not in _maps:
        _maps[key] = ops.pack_map(net, False).to(device)
    flats = [ops.flatten_params(sd, prefix, net, device) for prefix in ("model_coarse.", "model_fine.")]
    blobs = [ops.pack_apply(_maps[key], flat) for flat in flats]
    packed = PackedNeRF(net, blobs[0], blobs[1])
    packed._flat = (flats[0], flats[1])      # the bf16 variant is packed from these, lazily, on the device
    packed._range_token = (model, tuple((p.data_ptr(), p._version) for p in model.parameters()))
    return packed
