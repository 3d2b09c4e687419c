"""The training step with the fp32 forward and with the split-precision forward, interleaved in one process on one box:
    python tools/train_f16s_probe.py [rays] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import numpy as np
import torch
from nerf_pytorch_paeng_amd import nerf_process as NP
from nerf_pytorch_paeng_amd import ops, synthetic
from nerf_pytorch_paeng_amd.model import NeRF, get_positional_encoder

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
sd = synthetic.make_state_dict(0, 8, 256)
K, H, W = synthetic.lego_camera()
pix = torch.from_numpy(synthetic.pixel_batch(H, W, n, 0)).to(dev)
o, d = ops.make_o_d_pixels(W, H, K, synthetic.pose_spherical(0.0, -30.0, 4.0), pix)
target = torch.rand(n, 3, device=dev)
opts = SimpleNamespace(near=2.0, far=6.0, N_samples_c=64, N_samples_f=128, perturb=1.0, chunk_rays=n, chunk_pts=524288, data_type="blender", gpu_ids=[0], rank=0)
posenc = get_positional_encoder(10), get_positional_encoder(4)
res = {False: [], True: []}
models = {}
for mode in (False, True):
    m = NeRF(8, 256, 63, 27).to(dev)
    m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    models[mode] = (m, torch.optim.Adam(m.parameters(), lr=5e-4))


def step(mode):
    m, opt = models[mode]
    rgb_c, _, rgb_f, _ = NP.batchify_rays_and_render_by_chunk(o, d, m, posenc, H, W, K, opts, seed=1, f16s=mode)
    opt.zero_grad()
    loss = ((rgb_c - target) ** 2).mean() + ((rgb_f - target) ** 2).mean()
    loss.backward()
    opt.step()
    return loss


for rnd in range(3):
    for mode in (False, True):
        for _ in range(2):
            step(mode)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step(mode)
        torch.cuda.synchronize()
        res[mode].append(1e3 * (time.perf_counter() - t0) / steps)
for mode in (False, True):
    print(f"{n} rays, forward {'f16 split' if mode else 'fp32 MFMA':10s}: {np.median(res[mode]):7.3f} ms per step  {[round(x, 3) for x in res[mode]]}  loss {float(loss):.5f}")
