"""NeRF coarse+fine MLP pair -- drop-in for the reference's model/NeRF.py:10-78.

Same constructor arguments, same sub-module and parameter names (so ``load_state_dict`` accepts the
reference's checkpoints, train.py:105-114), same ``forward(x, is_fine)`` contract.  The forward pass
is the hand-written MFMA kernel (``mi_nerf_mlp_embedded``): parameters are packed into the kernel's
streaming layout from their current values on every call (two gather launches).  Under ``torch.no_grad()`` (test.py:36,140) ``forward`` is
the inference kernel; with gradients enabled it is differentiable w.r.t. the selected network's parameters (hand-written
backward, train_path.py), so a caller that keeps the reference's own ``nerf_process.py`` and swaps only the model still trains.
"""
from __future__ import annotations

from typing import Sequence

import torch
import torch.nn as nn

from .. import ops
from .._lib import MiNerfError, as_f32_dev


class NeRFModule(nn.Module):
    def __init__(self, D: int, W: int, input_ch: int, input_ch_d: int, skips: Sequence[int] = (4,)):
        super().__init__()
        self.D, self.W = D, W
        self.input_ch_x, self.input_ch_d = input_ch, input_ch_d
        self.skips = list(skips)
        # NeRF.py:24-30 -- layer i+1 takes [gamma(x), h] iff i is a skip index
        self.linear_x = nn.ModuleList(
            [nn.Linear(input_ch, W)] + [nn.Linear(W + input_ch if i in self.skips else W, W) for i in range(D - 1)])
        self.linear_d = nn.Linear(input_ch_d + W, W // 2)
        self.linear_feat = nn.Linear(W, W)
        self.linear_density = nn.Linear(W, 1)
        self.linear_color = nn.Linear(W // 2, 3)

    def __getstate__(self):
        # the back reference to the parent NeRF is a weakref: not state (pickle / deepcopy / torch.save); NeRF re-binds it
        state = self.__dict__.copy()
        state.pop("_parent", None)
        return state

    def forward(self, x):
        """x [n, input_ch + input_ch_d] -> [n, 4], like the reference's sub-module (model/NeRF.py:33-52; its NeRF.forward merely
        dispatches to ``model_coarse`` / ``model_fine``, NeRF.py:75-78).  Routed through the parent NeRF, which owns the packed
        blobs of both networks."""
        parent = getattr(self, "_parent", None)
        parent = parent() if parent is not None else None
        if parent is None:
            raise MiNerfError("this NeRFModule is not part of a NeRF (model/NeRF.py:58-59): construct it through NeRF(...)")
        return parent(x, is_fine=parent.model_fine is self)


class NeRF(nn.Module):
    def __init__(self, D: int, W: int, input_ch: int, input_ch_d: int, skips: Sequence[int] = (4,), gt_camera_param=None, device=None):
        super().__init__()
        self.model_coarse = NeRFModule(D, W, input_ch, input_ch_d, skips)
        self.model_fine = NeRFModule(D, W, input_ch, input_ch_d, skips)
        self._bind_children()
        self.apply(self._init_weights)                                   # NeRF.py:60,63-65
        self.gt_intrinsic, self.gt_extrinsic = gt_camera_param if gt_camera_param is not None else (None, None)

    def _bind_children(self) -> None:
        """Weak back references (not sub-modules, not state): ``model.model_coarse(x)`` works like the reference's.  Re-made after
        unpickling / ``copy.deepcopy`` (``__setstate__``), so a copy's sub-modules route through the COPY."""
        import weakref
        for child in (self.model_coarse, self.model_fine):
            object.__setattr__(child, "_parent", weakref.ref(self))

    def __setstate__(self, state):
        super().__setstate__(state)
        self._bind_children()

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.xavier_uniform_(m.weight)

    def get_camera_gt(self):
        return self.gt_intrinsic, self.gt_extrinsic

    def forward(self, x, is_fine: bool = False):
        """x [n, input_ch + input_ch_d] -> [n, 4] = cat([rgb_raw, density_raw]) (NeRF.py:51,70-78)."""
        from ..weights import packed_for
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # the reference's own render_rays calls model(embedded, is_fine) with gradients enabled (nerf_process.py:190-192,
            # 206-207): differentiable w.r.t. the selected network's parameters (nothing flows into x)
            from .. import train_path
            st = train_path._state_for(self)
            x = as_f32_dev(x, st.device)
            lead = x.shape[:-1]
            params = st.params(self.model_fine if is_fine else self.model_coarse)
            out = train_path._EmbeddedTrain.apply(st, x.reshape(-1, x.shape[-1]).contiguous(), *params)
            return out.reshape(*lead, 4)
        packed = packed_for(self)
        x = as_f32_dev(x, packed.device)
        lead = x.shape[:-1]
        out = ops.mlp_embedded(packed.net, packed.blob(is_fine), x.reshape(-1, x.shape[-1]))
        return out.reshape(*lead, 4)
