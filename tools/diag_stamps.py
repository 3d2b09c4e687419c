#!/usr/bin/env python3
"""Run the fused MLP once through the MN_DIAG build (build_scratch/libmi_nerf_diag.so) and print where a tile's cycles go.
Diagnostic only: the stamped build serialises around every stamp; read its SHARES, never its run time."""
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
pix = torch.from_numpy(synthetic.pixel_batch(H, W, 4096, 0)).to(dev)
o, d = ops.make_o_d_pixels(W, H, K, synthetic.pose_spherical(0.0, -30.0, 4.0), pix)
rays = torch.cat([o, d], -1).contiguous()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 192
z = torch.sort(torch.rand(4096, S, device=dev) * 4 + 2, -1)[0]
for _ in range(3):
    ops.mlp_rays(packed.net, packed.fine, rays, z)
torch.cuda.synchronize()
print("ideal MFMA cycles per tile: layer0 16384, trunk 475136, feature 65536, viewdir 32768, total 589824")
if len(sys.argv) > 2 and sys.argv[2] == "stash":
    print("---- training forward (STASH) ----")
    ops.mlp_rays_train(packed.net, packed.fine, rays, z)
    torch.cuda.synchronize()
