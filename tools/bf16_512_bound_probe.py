"""Round 5, VERDICT item 3: what is there to win on the bf16 step at the shape ONE OF EIGHT ranks runs for BASELINE config #5 (512 rays, 64 + 128
samples)?  A timing-only bound, measured before anything is built (the form profiles/r04_wgrad_f16s_nodup_bound.txt took):

  step        the shipped render_rays step (4 dependent launches: coarse net [depths drawn inside] | composite + resample + merge | fine net | composite)
  nets        the SAME two network launches back to back with the depths pre-computed and NO stage launch between them: the step with the
              stage work and two of its three launch boundaries at zero cost -- the upper bound of fusing the composites into the network launches
  launches    each network launch alone (hipEvents inside the library), in the launcher's own plan and with the 64- / 32-point shape pinned:
              the pass-time model (a pass costs the same however many CUs it fills)
  stage       composite_fine_z + composite alone (the fp32 stage kernels both modes share), back to back

    python tools/bf16_512_bound_probe.py [rays ...]    ->  profiles/r05_bf16_512_ray_bound.txt"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from nerf_pytorch_paeng_amd import ops, synthetic, weights
from nerf_pytorch_paeng_amd._lib import check, dev_ptr, lib, stream_ptr

dev = torch.device("cuda:0")
SC, NF = 64, 128
sd = synthetic.make_state_dict(0, 8, 256)
packed = weights.PackedNeRF.from_state_dict(sd, dev)
blobs = packed.bf16()
net = packed.net
K, H, W = synthetic.lego_camera()
pose = synthetic.pose_spherical(0.0, -30.0, 4.0)
rays_list = [int(a) for a in sys.argv[1:]] or [512, 256, 1024]
ROUNDS, REPS = 5, 400


PREWARM_S = 0.08     # the clock governor needs tens of milliseconds of the SAME load to settle (profiles/r04_prewarm_512_ray_shard.txt)


def wall(fn, reps=REPS):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < PREWARM_S:
        for _ in range(8):
            fn()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / reps


for n in rays_list:
    pix = torch.from_numpy(synthetic.pixel_batch(H, W, n, 0)).to(dev)
    o, d = ops.make_o_d_pixels(W, H, K, pose, pix)
    rays = torch.cat([o, d], -1).contiguous()
    cfg = ops.render_cfg(2.0, 6.0, SC, NF, False, True, seed=0)
    out = (torch.empty(n, 3, device=dev), torch.empty(n, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, device=dev))
    ws = torch.empty(ops.workspace_layout(cfg, n).total, dtype=torch.uint8, device=dev)
    ops.render_rays(net, blobs[0], blobs[1], cfg, rays, None, None, workspace=ws, out=out)
    v = ops.workspace_views(cfg, n, ws)
    z_c, z_f, raw_c0 = v["z_c"].clone(), v["z_f"].clone(), v["raw_c"].clone()
    raw_c, raw_f = torch.empty(n, SC, 4, device=dev), torch.empty(n, SC + NF, 4, device=dev)
    L, sp = lib(), stream_ptr(dev)
    a_c = (C.byref(net), dev_ptr(blobs[0], "p", torch.uint8, 16), dev_ptr(rays), dev_ptr(z_c), n, SC, dev_ptr(raw_c, align=16), 0, sp)
    a_f = (C.byref(net), dev_ptr(blobs[1], "p", torch.uint8, 16), dev_ptr(rays), dev_ptr(z_f), n, SC + NF, dev_ptr(raw_f, align=16), 0, sp)

    def step():
        ops.render_rays(net, blobs[0], blobs[1], cfg, rays, None, None, workspace=ws, out=out)

    def nets():
        check(L.mi_nerf_mlp_rays_bf16_shape(*a_c), "coarse")
        check(L.mi_nerf_mlp_rays_bf16_shape(*a_f), "fine")

    w_c = v["weights_c"].clone()
    u = ops.fill_uniform(0, 1, 0, n, NF, dev)

    def stage():                                              # the two stage launches of the step, as separate entry points (composite + fine_z | composite)
        ops.composite(raw_c0, z_c, rays, want_all=True)
        ops.fine_z(z_c, w_c, NF, False, u)
        ops.composite(raw_f, z_f, rays, want_all=False)

    res = {"step": [], "nets": [], "stage": [], "coarse": {0: [], 64: [], 32: []}, "fine": {0: [], 64: [], 32: []}}
    for _ in range(ROUNDS):
        res["step"].append(wall(step))
        res["nets"].append(wall(nets))
        res["stage"].append(wall(stage))
        for s in (0, 64, 32):
            ops.time_mlp_rays(net, blobs[0], rays, z_c, raw_c, 1500, True, s)          # ~60 ms of the same launch first
            res["coarse"][s].append(1e3 * ops.time_mlp_rays(net, blobs[0], rays, z_c, raw_c, 200, True, s))
            ops.time_mlp_rays(net, blobs[1], rays, z_f, raw_f, 700, True, s)
            res["fine"][s].append(1e3 * ops.time_mlp_rays(net, blobs[1], rays, z_f, raw_f, 200, True, s))
    med = lambda x: float(np.median(x))
    st, nt = med(res["step"]), med(res["nets"])
    c0, f0 = med(res["coarse"][0]), med(res["fine"][0])
    print(f"{n:5d} rays  step {st:7.1f} us | the two network launches back to back {nt:7.1f} us (coarse alone {c0:6.1f} + fine alone {f0:6.1f} = {c0 + f0:6.1f}) | "
          f"stage work + 2 launch boundaries = {st - nt:5.1f} us = {100 * (st - nt) / st:4.1f} % of the step: the UPPER BOUND of fusing the composites into the network launches", flush=True)
    print(f"            pinned shapes: coarse 64-pt {med(res['coarse'][64]):6.1f} / 32-pt {med(res['coarse'][32]):6.1f} us ({n * SC} points = {n * SC / 65536:.2f} chip passes of 64 points per wave); "
          f"fine 64-pt {med(res['fine'][64]):6.1f} / 32-pt {med(res['fine'][32]):6.1f} us ({n * (SC + NF)} points = {n * (SC + NF) / 65536:.2f} passes)", flush=True)
    print(f"            the stage work as three separate launches (composite | sample_pdf + merge | composite), back to back: {med(res['stage']):5.1f} us", flush=True)
    flop = n * (2 * SC + NF) * 1186816
    print(f"            at the 4096-ray launch's own rate (0.68 of 2.5 PFLOP/s) the two nets would take {flop / (0.68 * 2.5e15) * 1e6:6.1f} us; they take {c0 + f0:6.1f}: "
          f"the difference is the half-filled pass structure (a pass of the weight stream costs the same however many CUs it fills)", flush=True)
