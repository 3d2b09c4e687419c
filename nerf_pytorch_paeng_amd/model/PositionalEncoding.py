"""Positional encoding -- drop-in for the reference's model/PositionalEncoding.py:7-36.

``get_positional_encoder(L)`` returns ``(fn, out_dim)`` like the reference; ``fn`` maps an
``[n, 3]`` fp32 device tensor to ``[n, 3 + 6L]`` = ``[x, sin(2^0 x), cos(2^0 x), ...]`` through the
HIP kernel (``mi_nerf_posenc``).  The closure carries ``fn.L`` so the fused render path can recover
the frequency count without calling it (closures cannot cross the C ABI).
"""
from __future__ import annotations

import torch

from .. import ops
from .._lib import as_f32_dev


class PositionalEncoding:
    def __init__(self, L: int):
        self.L = int(L)
        self.out_dim = 3 + 6 * self.L                      # PositionalEncoding.py:13,24

    def embed(self, inputs: torch.Tensor) -> torch.Tensor:  # PositionalEncoding.py:29-30
        x = as_f32_dev(inputs)
        lead = x.shape[:-1]
        out = ops.posenc(x.reshape(-1, 3), self.L)
        return out.reshape(*lead, self.out_dim)


def get_positional_encoder(L: int):                         # PositionalEncoding.py:33-36
    obj = PositionalEncoding(L)

    def pos_encoder(x, eo=obj):
        return eo.embed(x)

    pos_encoder.L = obj.L
    pos_encoder.out_dim = obj.out_dim
    return pos_encoder, obj.out_dim
