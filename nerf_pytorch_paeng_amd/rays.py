"""Ray generation -- drop-in for the reference's rays.py:7-34.

``make_o_d`` runs the HIP ray-generation kernel on the pose's device; ``get_rays_np`` returns host
numpy arrays like the reference's numpy twin (used for the global-batch precompute, main.py:95-103),
computed by the same kernel.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from ._lib import MiNerfError


def make_o_d(img_w: int, img_h: int, img_k, pose: torch.Tensor):
    """rays_o, rays_d [H, W, 3] fp32 on ``pose.device`` (rays.py:20-34).  ``rays_o`` is a stride-0
    expanded view of the camera centre, exactly like the reference (rays.py:33)."""
    if not isinstance(pose, torch.Tensor) or not pose.is_cuda:
        raise MiNerfError("pose must be a tensor on a HIP device (the reference takes the device from pose, rays.py:23)")
    _, d = ops.make_o_d(int(img_w), int(img_h), img_k, pose, pose.device, want_origins=False)
    o = pose[:3, -1].to(torch.float32).expand(d.shape)
    return o, d


def get_rays(H: int, W: int, K, c2w: torch.Tensor):
    """north-star alias (original NeRF naming): same as make_o_d with (H, W) argument order."""
    return make_o_d(W, H, K, c2w)


def get_rays_np(H: int, W: int, K, c2w, device=None):
    """numpy rays_o, rays_d [H, W, 3] (rays.py:7-17), float32 (the caller casts to float32 at main.py:101)."""
    device = device or (c2w.device if isinstance(c2w, torch.Tensor) and c2w.is_cuda else torch.device("cuda", torch.cuda.current_device()))
    pose = torch.as_tensor(np.asarray(c2w.detach().cpu() if isinstance(c2w, torch.Tensor) else c2w), dtype=torch.float32)
    _, d = ops.make_o_d(int(W), int(H), K, pose, device, want_origins=False)
    d_np = d.cpu().numpy()
    o_np = np.broadcast_to(pose.numpy()[:3, -1], d_np.shape)
    return o_np, d_np
