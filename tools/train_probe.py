"""Time the training step (forward + backward + Adam) of the drop-in path on one GPU: python tools/train_probe.py [rays] [steps]."""
import sys
import time
from types import SimpleNamespace

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nerf_pytorch_paeng_amd import nerf_process as NP, ops, synthetic
from nerf_pytorch_paeng_amd.model import NeRF, get_positional_encoder

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
sd = synthetic.make_state_dict(0, 8, 256)
model = NeRF(8, 256, 63, 27).to(dev)
model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
posenc = get_positional_encoder(10), get_positional_encoder(4)
opts = SimpleNamespace(near=2.0, far=6.0, N_samples_c=64, N_samples_f=128, perturb=1.0, chunk_rays=4096, chunk_pts=524288,
                       data_type="blender", gpu_ids=[0], rank=0)
K, H, W = synthetic.lego_camera()
pose = synthetic.pose_spherical(30.0, -30.0, 4.0)
pix = torch.from_numpy(synthetic.pixel_batch(H, W, n, 1)).to(dev)
o, d = ops.make_o_d_pixels(W, H, K, pose, pix)
target = torch.rand(n, 3, device=dev)
optim = torch.optim.Adam(model.parameters(), lr=5e-4, betas=(0.9, 0.999))


def step():
    rgb_c, _, rgb_f, _ = NP.batchify_rays_and_render_by_chunk(o, d, model, posenc, H, W, K, opts)
    optim.zero_grad()
    loss = torch.nn.functional.mse_loss(rgb_c, target) + torch.nn.functional.mse_loss(rgb_f, target)
    loss.backward()
    optim.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"train step: {dt * 1e3:.3f} ms  ({n / dt:.0f} rays/s)  loss {loss.item():.5f}  peak mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
