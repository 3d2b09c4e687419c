"""GPU debug aid for mlp_bf16.hip: the kernel against the bf16 oracle (oracle/restate.py mlp_forward_bf16) for several network shapes,
error split by output channel and by point tile of the pair.  python tools/bf16_debug.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from nerf_pytorch_paeng_amd import ops, synthetic, weights
from oracle import restate as R

dev = torch.device("cuda:0")
K, H, W = synthetic.lego_camera()
o, d = R.make_o_d(W, H, K, torch.from_numpy(synthetic.pose_spherical(0., -30., 4.)[:3, :4]))
pix = synthetic.pixel_batch(H, W, 4096, 0)
for (D, skip, n, S) in ((2, -1, 8, 64), (3, -1, 8, 64), (3, 0, 8, 64), (4, 1, 8, 64), (8, 4, 8, 64), (8, 4, 64, 192), (8, 4, 33, 100), (8, 4, 1500, 192)):
    sd = synthetic.make_state_dict(3, D, 256, skips=(skip,) if skip >= 0 else ())
    packed = weights.PackedNeRF.from_state_dict(sd, dev)
    rays = torch.cat([o.reshape(-1, 3)[pix[:n]], d.reshape(-1, 3)[pix[:n]]], -1).contiguous()
    z = torch.sort(torch.from_numpy(R.counter_uniform(2, 0, 0, n, S)) * 4 + 2, -1)[0]
    raw = ops.mlp_rays(packed.net, packed.bf16()[1], rays.to(dev), z.to(dev), bf16=True).cpu()
    torch.cuda.synchronize()
    x = R.embed(rays, z, 10, 4)
    ref = R.mlp_forward_bf16(sd, "model_fine.", x, D, 63, 27, skips=(skip,) if skip >= 0 else ()).reshape(n, S, 4)
    f32 = R.mlp_forward(sd, "model_fine.", x, D, 63, 27, skips=(skip,) if skip >= 0 else ()).reshape(n, S, 4)
    e = (raw - ref).abs()
    scale = ref.abs().mean((0, 1))
    tile = (torch.arange(S) // 32) % 2
    msg = f"D={D} skip={skip} n={n} S={S}: rel err vs bf16 oracle per channel {[round(float(e[..., c].mean() / scale[c]), 5) for c in range(4)]}"
    msg += f" | p0 {float(e[:, tile == 0].mean()):.2e} p1 {float(e[:, tile == 1].mean()):.2e} | oracle bf16 vs fp32 {float((ref - f32).abs().mean() / f32.abs().mean()):.2e}"
    msg += f" | finite {bool(torch.isfinite(raw).all())} max {float(e.max()):.3e}"
    print(msg, flush=True)
    if D == 2:
        lane = e[0, :64].mean(-1)
        print("   per-sample err (ray 0):", [round(float(v), 4) for v in lane[:64]])
