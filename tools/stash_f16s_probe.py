"""Fine-network training-forward launch (4096 x 192 points), fp32 STASH kernel and split-precision STASH kernel, hipEvent-timed through torch:
    [MI_NERF_LIB=variant.so] python tools/stash_f16s_probe.py [label]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_pytorch_paeng_amd import ops, synthetic, weights
dev = torch.device("cuda:0")
sd = synthetic.make_state_dict(0, 8, 256)
net = weights.infer_net(sd)
K, H, W = synthetic.lego_camera()
n, S = 4096, 192
pix = torch.from_numpy(synthetic.pixel_batch(H, W, n, 0)).to(dev)
o, d = ops.make_o_d_pixels(W, H, K, synthetic.pose_spherical(0.0, -30.0, 4.0), pix)
rays = torch.cat([o, d], -1).contiguous()
z = torch.sort(torch.rand(n, S, device=dev) * 4 + 2, -1)[0]
b32 = ops.pack_module(sd, "model_fine.", net).to(dev)
b16 = ops.pack_module(sd, "model_fine.", net, f16s=True).to(dev)
stash = torch.empty(ops.train_layout(net, n, S).stash_bytes, dtype=torch.uint8, device=dev)
out = {}
for name, blob, f16s in (("fp32 STASH", b32, False), ("f16s STASH", b16, True)):
    for _ in range(3):
        ops.mlp_rays_train(net, blob, rays, z, stash=stash, f16s=f16s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.mlp_rays_train(net, blob, rays, z, stash=stash, f16s=f16s)
    e1.record(); torch.cuda.synchronize()
    out[name] = e0.elapsed_time(e1) / 10
print(f"{sys.argv[1] if len(sys.argv) > 1 else 'shipped':12s} " + "  ".join(f"{k}: {v:7.3f} ms" for k, v in out.items()), flush=True)
