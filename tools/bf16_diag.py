#!/usr/bin/env python3
"""Run the bf16 fused MLP once through the MN_DIAG build (build_scratch/libmi_nerf_diag.so) and print where a tile pair's cycles go.
Diagnostic only: read SHARES, never its run time.   python tools/bf16_diag.py [S] [points_per_wave] [rays]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from nerf_pytorch_paeng_amd import _lib  # noqa: E402

from nerf_pytorch_paeng_amd import build as _build
_lib.LIB_PATH = _build.build_diag_library()          # build_scratch/libmi_nerf_diag.so, built here when missing
from nerf_pytorch_paeng_amd import ops, synthetic, weights  # noqa: E402

dev = torch.device("cuda:0")
packed = weights.PackedNeRF.from_state_dict(synthetic.make_state_dict(0, 8, 256), dev)
K, H, W = synthetic.lego_camera()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 192
PPW = int(sys.argv[2]) if len(sys.argv) > 2 else 0
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
pix = torch.from_numpy(synthetic.pixel_batch(H, W, N, 0)).to(dev)
o, d = ops.make_o_d_pixels(W, H, K, synthetic.pose_spherical(0.0, -30.0, 4.0), pix)
rays = torch.cat([o, d], -1).contiguous()
z = torch.sort(torch.rand(N, S, device=dev) * 4 + 2, -1)[0]
for _ in range(3):
    ops.mlp_rays(packed.net, packed.bf16()[1], rays, z, bf16=True, points_per_wave=PPW)
torch.cuda.synchronize()
