"""Global-batch staging on its own (for rocprofv3 --pmc passes): rays_rgb for 100 800x800 images + one row shuffle, and the
per-frame ops of the eval harness (image_metrics, nanmax, to8b) on an 800x800 frame."""
import numpy as np
import torch

from nerf_pytorch_paeng_amd import ops, synthetic

dev = torch.device("cuda:0")
K, H, W = synthetic.lego_camera()
n_img = 100
imgs = torch.rand(n_img, H, W, 3, device=dev)
poses = torch.from_numpy(np.stack([synthetic.pose_spherical(3.6 * i - 180.0, -30.0, 4.0)[:3, :4] for i in range(n_img)], 0)).float().to(dev).contiguous()
perm = torch.randperm(n_img * H * W, device=dev)
for _ in range(2):
    rr = ops.rays_rgb(W, H, K, poses, imgs)
    rr2 = ops.permute_rows(rr, perm)
pred, tgt = torch.rand(H * W, 3, device=dev), torch.rand(H * W, 3, device=dev)
disp = torch.rand(H * W, device=dev) * 5
for _ in range(2):
    m = ops.image_metrics(pred, tgt)
    mx = ops.nanmax(disp)
    b1, b2 = ops.to8b(pred), ops.to8b(disp, mx)
torch.cuda.synchronize()
print("ok", float(m[0]), float(mx))
