"""Deterministic synthetic inputs: weights, cameras, ray batches.

There is no dataset and no trained checkpoint in the build environment, so benchmarks, tests and
the golden-fixture generator all draw their inputs from here.  Everything is numpy
``RandomState`` based so it is bit-identical on every machine (the fixtures under
``tests/golden`` never store weight blobs -- they are regenerated from a seed).

State-dict layout follows the reference checkpoint (model/NeRF.py:24-30,58-59; train.py:105-114):
``model_{coarse,fine}.linear_x.{i}.{weight,bias}``, ``linear_d``, ``linear_feat``,
``linear_density``, ``linear_color``; weights are ``[out, in]``.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Sequence, Tuple

import numpy as np

LEGO_CAMERA_ANGLE_X = 0.6911112070083618      # blender lego transforms json (not in the reference repo)


def layer_shapes(D: int, W: int, in_x: int, in_d: int, skips: Sequence[int] = (4,)) -> "OrderedDict[str, Tuple[int, int]]":
    """name -> (out, in) of one NeRFModule (model/NeRF.py:24-30)."""
    shapes: "OrderedDict[str, Tuple[int, int]]" = OrderedDict()
    shapes["linear_x.0"] = (W, in_x)
    for i in range(D - 1):                      # NeRF.py:25: layer i+1 takes W+in_x iff i in skips
        shapes[f"linear_x.{i + 1}"] = (W, W + in_x if i in skips else W)
    shapes["linear_d"] = (W // 2, in_d + W)
    shapes["linear_feat"] = (W, W)
    shapes["linear_density"] = (1, W)
    shapes["linear_color"] = (3, W // 2)
    return shapes


def make_state_dict(seed: int = 0, D: int = 8, W: int = 256, in_x: int = 63, in_d: int = 27,
                    skips: Sequence[int] = (4,), density_scale: float = 20.0) -> Dict[str, np.ndarray]:
    """Xavier-uniform weights (NeRF.py:63-65), PyTorch-default biases U(+-1/sqrt(fan_in)).

    ``linear_density.weight`` is multiplied by ``density_scale`` so that alpha is non-degenerate
    and the coarse pdf is peaked (branch coverage of sample_pdf).
    """
    rs = np.random.RandomState(seed)
    sd: Dict[str, np.ndarray] = OrderedDict()
    for net in ("model_coarse", "model_fine"):
        for name, (fo, fi) in layer_shapes(D, W, in_x, in_d, skips).items():
            a = math.sqrt(6.0 / (fi + fo))
            w = rs.uniform(-a, a, size=(fo, fi)).astype(np.float32)
            b = rs.uniform(-1.0 / math.sqrt(fi), 1.0 / math.sqrt(fi), size=(fo,)).astype(np.float32)
            if name == "linear_density":
                w = (w * np.float32(density_scale)).astype(np.float32)
            sd[f"{net}.{name}.weight"] = w
            sd[f"{net}.{name}.bias"] = b
    return sd


# ---------------------------------------------------------------------------------------------
# cameras
# ---------------------------------------------------------------------------------------------
def pose_spherical(theta_deg: float, phi_deg: float, radius: float) -> np.ndarray:
    """Camera-to-world 4x4 on a sphere, looking at the origin (dataset/render_pose.py:5-34):
    translate along +z by ``radius``, rotate about x by phi, about y by theta, then swap axes."""
    th, ph = math.radians(theta_deg), math.radians(phi_deg)
    trans = np.eye(4, dtype=np.float32); trans[2, 3] = radius
    rx = np.array([[1, 0, 0, 0], [0, math.cos(ph), -math.sin(ph), 0], [0, math.sin(ph), math.cos(ph), 0], [0, 0, 0, 1]], dtype=np.float32)
    ry = np.array([[math.cos(th), 0, -math.sin(th), 0], [0, 1, 0, 0], [math.sin(th), 0, math.cos(th), 0], [0, 0, 0, 1]], dtype=np.float32)
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.float32)
    return (flip @ (ry @ (rx @ trans))).astype(np.float32)


def lego_camera(H: int = 800, W: int = 800) -> Tuple[np.ndarray, int, int]:
    """K (float64, as dataset/load_blender.py:51-52,66-70 builds it), H, W."""
    focal = 0.5 * W / math.tan(0.5 * LEGO_CAMERA_ANGLE_X)
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]], dtype=np.float64)
    return K, H, W


def fern_camera(H: int = 378, W: int = 504, focal: float = 407.5) -> Tuple[np.ndarray, int, int]:
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]], dtype=np.float64)
    return K, H, W


def fern_pose() -> np.ndarray:
    """Identity-ish recentred forward-facing pose with a +-0.1 translation."""
    c2w = np.eye(4, dtype=np.float32)
    a = 0.05
    c2w[:3, :3] = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]], dtype=np.float32)
    c2w[:3, 3] = np.array([0.1, -0.1, 0.05], dtype=np.float32)
    return c2w


def pixel_batch(H: int, W: int, n: int = 4096, seed: int = 0) -> np.ndarray:
    """Flat pixel indices of the benchmark batch: RandomState(seed).choice(H*W, n, replace=False)."""
    return np.random.RandomState(seed).choice(H * W, n, replace=False).astype(np.int64)
