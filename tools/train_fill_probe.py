"""Which PyTorch ops launch the small fill / copy kernels inside a training step (torch.profiler, 3 steps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
from torch.profiler import ProfilerActivity, profile
from nerf_pytorch_paeng_amd import nerf_process as NP, ops, synthetic
from nerf_pytorch_paeng_amd.model import NeRF, get_positional_encoder

n = 1024
dev = torch.device("cuda:0")
sd = synthetic.make_state_dict(0, 8, 256)
model = NeRF(8, 256, 63, 27).to(dev)
model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
posenc = get_positional_encoder(10), get_positional_encoder(4)
opts = SimpleNamespace(near=2.0, far=6.0, N_samples_c=64, N_samples_f=128, perturb=1.0, chunk_rays=4096, chunk_pts=524288,
                       data_type="blender", gpu_ids=[0], rank=0)
K, H, W = synthetic.lego_camera()
pix = torch.from_numpy(synthetic.pixel_batch(H, W, n, 1)).to(dev)
o, d = ops.make_o_d_pixels(W, H, K, synthetic.pose_spherical(30.0, -30.0, 4.0), pix)
target = torch.rand(n, 3, device=dev)
optim = torch.optim.Adam(model.parameters(), lr=5e-4, betas=(0.9, 0.999))


def step():
    rgb_c, _, rgb_f, _ = NP.batchify_rays_and_render_by_chunk(o, d, model, posenc, H, W, K, opts)
    optim.zero_grad()
    loss = torch.nn.functional.mse_loss(rgb_c, target) + torch.nn.functional.mse_loss(rgb_f, target)
    loss.backward()
    optim.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.count)
print(f"{'count/step':>10}  name")
for e in rows[:45]:
    print(f"{e.count / 3:10.1f}  {e.key[:110]}")
