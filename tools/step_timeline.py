"""One render_rays step at small ray counts, fp32 and bf16: wall time per step, for launch-gap analysis.
   python tools/step_timeline.py [--bf16-only] [--jitter] [rays ...]   (under rocprofv3 --kernel-trace: tools/rocpd_summary.py DB --timeline 16)
   --jitter: draw t_rand / u inside the step, as bench.py's step does."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
from nerf_pytorch_paeng_amd import ops, synthetic, weights

dev = torch.device("cuda:0")
SC, NF = 64, 128
sd = synthetic.make_state_dict(0, 8, 256)
packed = weights.PackedNeRF.from_state_dict(sd, dev)
K, H, W = synthetic.lego_camera()
pose = synthetic.pose_spherical(0.0, -30.0, 4.0)
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
JITTER, BF16_ONLY = "--jitter" in sys.argv, "--bf16-only" in sys.argv
rays_list = [int(a) for a in argv] or [512, 4096]
for bf16 in ((True,) if BF16_ONLY else (False, True)):
    cfg = ops.render_cfg(2.0, 6.0, SC, NF, False, bf16)
    blobs = packed.bf16() if bf16 else (packed.coarse, packed.fine)
    for n in rays_list:
        pix = torch.from_numpy(synthetic.pixel_batch(H, W, n, 0)).to(dev)
        o, d = ops.make_o_d_pixels(W, H, K, pose, pix)
        rays = torch.cat([o, d], -1).contiguous()
        t_rand, u = ops.fill_uniform(0, 0, 0, n, SC, dev), ops.fill_uniform(0, 1, 0, n, NF, dev)
        out = (torch.empty(n, 3, device=dev), torch.empty(n, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, device=dev))
        ws = torch.empty(ops.workspace_layout(cfg, n).total, dtype=torch.uint8, device=dev)
        def step():
            if JITTER:       # drawn by the sampling kernels themselves (seed 0, global ray index, sample), as bench.py's step does
                ops.render_rays(packed.net, blobs[0], blobs[1], cfg, rays, None, None, workspace=ws, out=out)
            else:
                ops.render_rays(packed.net, blobs[0], blobs[1], cfg, rays, t_rand, u, workspace=ws, out=out)
        t_pw = time.perf_counter()
        while time.perf_counter() - t_pw < 0.1:          # let the clock governor settle on this load before anything is read
            for _ in range(8):
                step()
            torch.cuda.synchronize()
        reps = 200
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / reps
        print(f"{'bf16' if bf16 else 'fp32'} {n:5d} rays: {ms:.4f} ms per step  {n / ms * 1e3:.0f} rays/s", flush=True)
