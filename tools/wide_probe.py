"""mlp_fp32_wide_kernel (netWidth 512, 16 points per wave on v_mfma_f32_16x16x4_f32): fine-launch time and fraction of the fp32 MFMA peak.
    python tools/wide_probe.py [rays]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_pytorch_paeng_amd import ops, synthetic, weights

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = 192
for W in (256, 512, 384, 320):
    sd = synthetic.make_state_dict(0, 8, W)
    packed = weights.PackedNeRF.from_state_dict(sd, dev)
    K, H, Wd = synthetic.lego_camera()
    pix = torch.from_numpy(synthetic.pixel_batch(H, Wd, n, 0)).to(dev)
    o, d = ops.make_o_d_pixels(Wd, H, K, synthetic.pose_spherical(0.0, -30.0, 4.0), pix)
    rays = torch.cat([o, d], -1).contiguous()
    z = torch.sort(torch.rand(n, S, device=dev) * 4 + 2, -1)[0]
    raw = torch.empty(n, S, 4, device=dev)
    macs = 63 * W + 4 * W * W + (W + 63) * W + 2 * W * W + W * W + W + (W + 27) * (W // 2) + 3 * (W // 2)
    ops.time_mlp_rays(packed.net, packed.fine, rays, z, raw, 3)
    ms = ops.time_mlp_rays(packed.net, packed.fine, rays, z, raw, 10)
    tf = 2 * macs * n * S / (ms * 1e-3) / 1e12
    print(f"W={W}: fine launch ({n * S} points) {ms:.3f} ms = {tf:.1f} TFLOP/s algorithmic = {tf / 157.3:.3f} of the fp32 MFMA peak ({2 * macs} FLOP/point)", flush=True)
