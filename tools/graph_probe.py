"""Does replaying one render_rays step from a captured HIP graph shorten it?  (torch.cuda.CUDAGraph around the C-ABI call on torch's stream)
   python tools/graph_probe.py [rays ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_pytorch_paeng_amd import ops, synthetic, weights

dev = torch.device("cuda:0")
SC, NF = 64, 128
sd = synthetic.make_state_dict(0, 8, 256)
packed = weights.PackedNeRF.from_state_dict(sd, dev)
K, H, W = synthetic.lego_camera()
pose = synthetic.pose_spherical(0.0, -30.0, 4.0)
rays_list = [int(a) for a in sys.argv[1:]] or [512, 4096]
for bf16 in (False, True):
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
            ops.render_rays(packed.net, blobs[0], blobs[1], cfg, rays, t_rand, u, workspace=ws, out=out)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        reps = 100
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        eager = 1e3 * (time.perf_counter() - t0) / reps
        ref = [t.clone() for t in out]
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            step()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                step()
        torch.cuda.synchronize()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        graph = 1e3 * (time.perf_counter() - t0) / reps
        same = all(torch.equal(a, b) for a, b in zip(ref, out))
        print(f"{'bf16' if bf16 else 'fp32'} {n:5d} rays: eager {eager:.4f} ms, graph replay {graph:.4f} ms ({100 * (eager - graph) / eager:+.1f} %), outputs identical: {same}", flush=True)
