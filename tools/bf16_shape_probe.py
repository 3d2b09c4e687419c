"""The bf16 kernel's two launch shapes (64 / 32 points per wave) side by side, one process, one box, interleaved rounds:
coarse launch (n x 64 points), fine launch (n x 192 points) and the whole render_rays step at several ray counts.
    python tools/bf16_shape_probe.py [rays ...]   ->  profiles/r03_bf16_small_launch_shape.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nerf_pytorch_paeng_amd import ops, synthetic, weights

dev = torch.device("cuda:0")
SC, NF = 64, 128
sd = synthetic.make_state_dict(0, 8, 256)
packed = weights.PackedNeRF.from_state_dict(sd, dev)
blobs = packed.bf16()
K, H, W = synthetic.lego_camera()
pose = synthetic.pose_spherical(0.0, -30.0, 4.0)
rays_list = [int(a) for a in sys.argv[1:]] or [128, 256, 384, 512, 768, 1024, 2048, 4096]
SHAPES = (64, 32, 0)          # (an 8-wave shape, 832, was measured and dropped: tools/ABLATIONS.md)
ROUNDS = 5
for n in rays_list:
    pix = torch.from_numpy(synthetic.pixel_batch(H, W, n, 0)).to(dev)
    o, d = ops.make_o_d_pixels(W, H, K, pose, pix)
    rays = torch.cat([o, d], -1).contiguous()
    t_rand, u = ops.fill_uniform(0, 0, 0, n, SC, dev), ops.fill_uniform(0, 1, 0, n, NF, dev)
    out = (torch.empty(n, 3, device=dev), torch.empty(n, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, device=dev))
    cfgs = {s: ops.render_cfg(2.0, 6.0, SC, NF, False, True, points_per_wave=s) for s in SHAPES}
    ws = torch.empty(ops.workspace_layout(cfgs[0], n).total, dtype=torch.uint8, device=dev)
    ops.render_rays(packed.net, blobs[0], blobs[1], cfgs[0], rays, t_rand, u, workspace=ws, out=out)
    v = ops.workspace_views(cfgs[0], n, ws)
    z_c, z_f = v["z_c"].clone(), v["z_f"].clone()
    raw_c, raw_f = torch.empty(n, SC, 4, device=dev), torch.empty(n, SC + NF, 4, device=dev)
    res = {s: {"coarse": [], "fine": [], "step": []} for s in SHAPES}
    for _ in range(ROUNDS):
        for s in SHAPES:
            ops.time_mlp_rays(packed.net, blobs[0], rays, z_c, raw_c, 3, True, s)
            res[s]["coarse"].append(ops.time_mlp_rays(packed.net, blobs[0], rays, z_c, raw_c, 20, True, s))
            ops.time_mlp_rays(packed.net, blobs[1], rays, z_f, raw_f, 3, True, s)
            res[s]["fine"].append(ops.time_mlp_rays(packed.net, blobs[1], rays, z_f, raw_f, 20, True, s))
            for _ in range(5):
                ops.render_rays(packed.net, blobs[0], blobs[1], cfgs[s], rays, t_rand, u, workspace=ws, out=out)
            torch.cuda.synchronize()
            reps = 40
            t0 = time.perf_counter()
            for _ in range(reps):
                ops.render_rays(packed.net, blobs[0], blobs[1], cfgs[s], rays, t_rand, u, workspace=ws, out=out)
            torch.cuda.synchronize()
            res[s]["step"].append(1e3 * (time.perf_counter() - t0) / reps)
    # the launcher's own plan with the jitter drawn inside the sampling kernels (what bench.py's step does)
    own = []
    for _ in range(ROUNDS):
        for _ in range(5):
            ops.render_rays(packed.net, blobs[0], blobs[1], cfgs[0], rays, None, None, workspace=ws, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            ops.render_rays(packed.net, blobs[0], blobs[1], cfgs[0], rays, None, None, workspace=ws, out=out)
        torch.cuda.synchronize()
        own.append(1e3 * (time.perf_counter() - t0) / 40)
    for s in SHAPES:
        c, f, st = (float(np.median(res[s][k])) for k in ("coarse", "fine", "step"))
        print(f"{n:5d} rays  {({0: 'auto', 832: '8x32'}.get(s, s))!s:>4} pts/wave: coarse launch {1e3 * c:7.1f} us  fine launch {1e3 * f:7.1f} us  step {1e3 * st:7.1f} us"
              f"  = {n / st * 1e3 / 1e6:6.3f} M rays/s", flush=True)
    st = float(np.median(own))
    print(f"{n:5d} rays  auto, jitter drawn in the kernels: step {1e3 * st:7.1f} us  = {n / st * 1e3 / 1e6:6.3f} M rays/s", flush=True)
