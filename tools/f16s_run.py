"""The split-precision step on its own (for rocprofv3: kernel stats and PMC passes of mlp_f16s_kernel): BASELINE config #2's batch, a few warm-up
steps, then `steps` steps and `launches` extra fine-network launches.   python3 tools/f16s_run.py [rays] [steps] [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_pytorch_paeng_amd import ops, synthetic, weights

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda:0")
SC, NF = 64, 128
packed = weights.PackedNeRF.from_state_dict(synthetic.make_state_dict(0, 8, 256), dev)
bc, bf = packed.f16s()
K, H, W = synthetic.lego_camera()
pix = torch.from_numpy(synthetic.pixel_batch(H, W, n, 0)).to(dev)
o, d = ops.make_o_d_pixels(W, H, K, synthetic.pose_spherical(0.0, -30.0, 4.0), pix)
rays = torch.cat([o, d], -1).contiguous()
cfg = ops.render_cfg(2.0, 6.0, SC, NF, False, seed=0, ray_offset=0, f16s=True)
ws = torch.empty(ops.workspace_layout(cfg, n).total, dtype=torch.uint8, device=dev)
out = (torch.empty(n, 3, device=dev), torch.empty(n, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, device=dev))
for _ in range(3 + steps):
    ops.render_rays(packed.net, bc, bf, cfg, rays, None, None, workspace=ws, out=out)
torch.cuda.synchronize()
z_f = ops.workspace_views(cfg, n, ws)["z_f"].clone()
raw = torch.empty(n, SC + NF, 4, device=dev)
ms = ops.time_mlp_rays(packed.net, bf, rays, z_f, raw, launches, f16s=True)
print(f"{n} rays: fine launch {ms:.4f} ms (hipEvents, {launches} launches)  finite {bool(torch.isfinite(out[2]).all())}")
