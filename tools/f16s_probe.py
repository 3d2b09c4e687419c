"""The three precision modes of the fused MLP side by side (one process, one box, interleaved rounds): fp32 MFMA, f16 split precision
(fp32-grade results on the f16 matrix pipe), bf16 -- fine launch, coarse launch and the whole render_rays step.
    python tools/f16s_probe.py [rays ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nerf_pytorch_paeng_amd import ops, synthetic, weights

dev = torch.device("cuda:0")
SC, NF = 64, 128
FLOP_PT = 2 * 593408
packed = weights.PackedNeRF.from_state_dict(synthetic.make_state_dict(0, 8, 256), dev)
K, H, W = synthetic.lego_camera()
pose = synthetic.pose_spherical(0.0, -30.0, 4.0)
modes = {"fp32": dict(blobs=(packed.coarse, packed.fine), kw={}), "f16s": dict(blobs=packed.f16s(), kw=dict(f16s=True)),
         "bf16": dict(blobs=packed.bf16(), kw=dict(bf16=True))}
for n in [int(a) for a in sys.argv[1:]] or [512, 4096]:
    pix = torch.from_numpy(synthetic.pixel_batch(H, W, n, 0)).to(dev)
    o, d = ops.make_o_d_pixels(W, H, K, pose, pix)
    rays = torch.cat([o, d], -1).contiguous()
    out = (torch.empty(n, 3, device=dev), torch.empty(n, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, device=dev))
    cfg0 = ops.render_cfg(2.0, 6.0, SC, NF, False)
    ws = torch.empty(ops.workspace_layout(cfg0, n).total, dtype=torch.uint8, device=dev)
    ops.render_rays(packed.net, packed.coarse, packed.fine, cfg0, rays, None, None, workspace=ws, out=out)
    v = ops.workspace_views(cfg0, n, ws)
    z_c, z_f = v["z_c"].clone(), v["z_f"].clone()
    raw_c, raw_f = torch.empty(n, SC, 4, device=dev), torch.empty(n, SC + NF, 4, device=dev)
    res = {m: {"coarse": [], "fine": [], "step": []} for m in modes}
    for _ in range(4):
        for m, spec in modes.items():
            cfg = ops.render_cfg(2.0, 6.0, SC, NF, False, **spec["kw"])
            bc, bf = spec["blobs"]
            ops.time_mlp_rays(packed.net, bc, rays, z_c, raw_c, 2, **spec["kw"])
            res[m]["coarse"].append(ops.time_mlp_rays(packed.net, bc, rays, z_c, raw_c, 10, **spec["kw"]))
            ops.time_mlp_rays(packed.net, bf, rays, z_f, raw_f, 2, **spec["kw"])
            res[m]["fine"].append(ops.time_mlp_rays(packed.net, bf, rays, z_f, raw_f, 10, **spec["kw"]))
            for _ in range(3):
                ops.render_rays(packed.net, bc, bf, cfg, rays, None, None, workspace=ws, out=out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                ops.render_rays(packed.net, bc, bf, cfg, rays, None, None, workspace=ws, out=out)
            torch.cuda.synchronize()
            res[m]["step"].append(1e3 * (time.perf_counter() - t0) / 20)
    for m in modes:
        c, f, st = (float(np.median(res[m][k])) for k in ("coarse", "fine", "step"))
        tf = n * (SC + NF) * FLOP_PT / (f * 1e-3) / 1e12
        print(f"{n:5d} rays  {m}: coarse launch {c:8.4f} ms  fine launch {f:8.4f} ms = {tf:7.1f} TFLOP/s of network arithmetic "
              f"({tf / 157.3:5.2f} x the fp32 MFMA peak)   step {st:8.4f} ms = {n / st * 1e3 / 1e3:9.1f} k rays/s", flush=True)
