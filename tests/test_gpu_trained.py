"""Parity and PSNR on TRAINED weights (the north star's own bar: "outputs match the reference PyTorch path on the same rays --
PSNR within 0.05 dB, per-pixel RGB within 1e-4 fp32"; reference callers test.py:17-72, utils.py:18-23).

Every other parity test uses `synthetic.make_state_dict` (Xavier + density x 20) or a freshly initialised net.  What test.py:20-21
loads is a TRAINED network: larger weights, sharp densities, saturated colours.  No dataset or checkpoint exists on the GPU box, so
this module makes the nearest thing itself: the BASELINE network (D = 8, W = 256, skip 4, 64 + 128 samples) is trained with
`harness.train` (Adam, 1024-ray steps of the global batch, lego camera geometry at 48 x 48) on three scenes

* ``teacher`` -- images of a fixed random 8 x 256 NeRF rendered by the inference kernels (the scene of test_gpu_harness.py), and
* ``solids``  -- an analytic scene with hard surfaces in front of the white background (a textured sphere, a box, a chequered slab;
                 ground truth ray-marched in float64 by plain torch), which is what drives a NeRF's weights to lego-like magnitudes, and
* ``plumbing`` -- BASELINE config #1: the 4 x 128 network, coarse only (N_samples_f = 0), on the solids scene, and
* ``teacher_llff`` -- BASELINE config #4's path: a forward-facing rig of fern-like cameras (36 x 48), rays through the NDC warp, near / far 0 / 1,
                 ``data_type = 'llff'`` (nerf_process.py:224-226), images of a random teacher rendered the same way,

saved and re-loaded through the reference's checkpoint format, and then compared with the pinned CPU oracle ON THE TRAINED WEIGHTS:

(i)   `render_rays` on 1024 full-resolution lego rays with injected randoms: `rgb_c`, depth-pinned `rgb_f`, share of un-pinned rays
      beyond 1e-4 (sample_pdf is discontinuous: SURVEY.md section 7);
(ii)  the held-out frame through `harness.test`: |PSNR_HIP - PSNR_oracle| < 0.05 dB against the scene's own image;
(iii) the same frame in bf16 and f16s: PSNR against the fp32 frame and the change of the PSNR against ground truth;
(iv)  the largest |pre-activation| of the trained nets against the f16 range (65 504) the split-precision mode has to live in.

The figures printed here are recorded in DESIGN.md section 2."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from nerf_pytorch_paeng_amd import harness, ops, synthetic, weights
from nerf_pytorch_paeng_amd import nerf_process as NP
from nerf_pytorch_paeng_amd.model import NeRF, get_positional_encoder
from oracle import restate as R

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
D, WD, SC, NF = 8, 256, 64, 128
# defaults sized for the GPU suite (~22 s per scene: the suite has a 300 s budget); TRAINED_STEPS / TRAINED_VIEWS / TRAINED_SIZE / TRAINED_SCENES scale the
# same tests up for a one-off run on longer-trained weights (profiles/r05_trained_weights_long.txt: 20 000 steps, 24 views of 96 x 96; rounds 5-6 ran 4000 steps
# by default: profiles/r05_trained_weights.txt, r06_trained_weights.txt)
N_IMG, N_STEPS, N_RAYS = int(os.environ.get("TRAINED_VIEWS", 8)), int(os.environ.get("TRAINED_STEPS", 3000)), 1024
SIZE = int(os.environ.get("TRAINED_SIZE", 48))
SCENES = tuple(os.environ.get("TRAINED_SCENES", "teacher,solids,teacher_llff,plumbing").split(","))


def _geo(scene):
    """Frame size, full-resolution camera, depth range and data type of a scene."""
    if scene.endswith("_llff"):
        Kf, Hf, Wf = synthetic.fern_camera()
        return SimpleNamespace(llff=True, HS=SIZE * 3 // 4, WS=SIZE, Kf=Kf, Hf=Hf, Wf=Wf, near=0.0, far=1.0, data_type="llff", D=D, WD=WD, NF=NF)
    Kf, Hf, Wf = synthetic.lego_camera()
    if scene == "plumbing":          # BASELINE config #1: 4 x 128, coarse only (N_samples_f = 0; with D = 4 the skip at 4 never fires), the solids scene
        return SimpleNamespace(llff=False, HS=SIZE, WS=SIZE, Kf=Kf, Hf=Hf, Wf=Wf, near=2.0, far=6.0, data_type="blender", D=4, WD=128, NF=0)
    return SimpleNamespace(llff=False, HS=SIZE, WS=SIZE, Kf=Kf, Hf=Hf, Wf=Wf, near=2.0, far=6.0, data_type="blender", D=D, WD=WD, NF=NF)


def _opts(geo, **kw):
    base = dict(near=geo.near, far=geo.far, N_samples_c=SC, N_samples_f=geo.NF, perturb=1.0, chunk_rays=4096, chunk_pts=524288, data_type=geo.data_type,
                gpu_ids=[0], rank=0, exp_name="trained", N_rays=N_RAYS, global_batch=True, idx_save=N_STEPS, n_angle=N_IMG + 1,
                single_angle=-1, phi=-30.0, nf=4.0, precision="fp32")
    base.update(kw)
    return SimpleNamespace(**base)


def _cfg(geo):
    return R.PathConfig(near=geo.near, far=geo.far, N_samples_c=SC, N_samples_f=geo.NF, perturb=1.0, netDepth=geo.D, netWidth=geo.WD, data_type=geo.data_type)


def _final(geo, out):
    """The colours the harness keeps: fine when N_samples_f > 0, else coarse (test.py:42-47)."""
    return out["rgb_f"] if geo.NF > 0 else out["rgb_c"]


def _llff_poses():
    """A forward-facing rig: the fern-like camera (synthetic.fern_pose) moved round a small circle in its image plane; the last one (held out) at the centre."""
    base = torch.from_numpy(synthetic.fern_pose()).float()
    poses = []
    for i in range(N_IMG + 1):
        p = base.clone()
        if i < N_IMG:
            th = 2.0 * np.pi * i / N_IMG
            p[:3, 3] += torch.tensor([0.15 * np.cos(th), 0.15 * np.sin(th), 0.03 * np.sin(2 * th)], dtype=torch.float32)
        poses.append(p)
    return torch.stack(poses, 0)


def _solids_images(K, poses, HS=48, WS=48):
    """Ground truth of the analytic scene: 1024 uniform depths per ray, float64, alpha-composited on white (plain torch, test-only)."""
    imgs = []
    for pose in poses:
        o, d = harness.make_o_d(WS, HS, K, pose[:3, :4].to(DEV))
        o, d = o.expand_as(d).reshape(-1, 3).double(), d.reshape(-1, 3).double()
        z = torch.linspace(2.0, 6.0, 1025, device=DEV, dtype=torch.float64)
        zm = 0.5 * (z[1:] + z[:-1])
        p = o[:, None, :] + d[:, None, :] * zm[None, :, None]                      # [n, 1024, 3]
        x, y, zc = p[..., 0], p[..., 1], p[..., 2]
        sph = ((x - 0.35) ** 2 + (y + 0.1) ** 2 + (zc - 0.15) ** 2) < 0.65 ** 2
        box = ((x + 0.75).abs() < 0.35) & ((y - 0.45).abs() < 0.35) & ((zc + 0.1).abs() < 0.5)
        slab = (x.abs() < 1.3) & (y.abs() < 1.3) & ((zc + 0.75).abs() < 0.06)
        sigma = 60.0 * (sph | box | slab).double()
        chk = ((torch.floor(x * 2.5) + torch.floor(y * 2.5)) % 2 == 0).double()
        col = torch.zeros(*p.shape[:2], 3, device=DEV, dtype=torch.float64)
        col[slab] = torch.stack([0.25 + 0.6 * chk, 0.25 + 0.6 * chk, 0.3 + 0.1 * chk], -1)[slab]
        col[box] = torch.stack([0.15 + 0 * x, 0.35 + 0.3 * torch.sin(9.0 * zc) ** 2, 0.85 + 0 * x], -1)[box]
        col[sph] = torch.stack([0.9 + 0 * x, 0.25 + 0.5 * torch.sin(7.0 * x + 3.0 * zc) ** 2, 0.15 + 0 * x], -1)[sph]
        dist = (z[1:] - z[:-1])[None, :] * d.norm(dim=-1, keepdim=True)
        alpha = 1.0 - torch.exp(-sigma * dist)
        T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha], -1), -1)[:, :-1]
        w = alpha * T
        rgb = (w[..., None] * col).sum(1) + (1.0 - w.sum(1, keepdim=True))
        imgs.append(rgb.float().reshape(HS, WS, 3))
    return torch.stack(imgs, 0)


@pytest.fixture(scope="module", params=SCENES)
def trained(request, tmp_path_factory):
    """Train once per scene; everything below reads the weights back from the reference-format checkpoint."""
    scene = request.param
    geo = _geo(scene)
    HS, WS = geo.HS, geo.WS
    tmp = str(tmp_path_factory.mktemp(f"ckpt_{scene}"))
    torch.manual_seed(11)
    NP.manual_seed(5)
    K = np.array([[geo.Kf[0][0] * WS / geo.Wf, 0, WS / 2], [0, geo.Kf[1][1] * HS / geo.Hf, HS / 2], [0, 0, 1]])
    posenc = get_positional_encoder(10), get_positional_encoder(4)
    poses = _llff_poses() if geo.llff else harness.get_render_pose(n_angle=N_IMG + 1, phi=-30.0, nf=4.0)
    opts = _opts(geo)
    if scene.startswith("teacher"):
        teacher = NeRF(D, WD, 63, 27).to(DEV)
        teacher.load_state_dict({k: torch.as_tensor(v) for k, v in synthetic.make_state_dict(77, D, WD).items()})
        with torch.no_grad():
            imgs = torch.stack([harness._render_pose(teacher, posenc, K, poses[i].to(DEV), (HS, WS), opts)[0].reshape(HS, WS, 3)
                                for i in range(N_IMG + 1)], 0)
    else:                                                            # "solids" and "plumbing": the analytic scene
        imgs = _solids_images(K, poses, HS, WS)
    assert float(imgs.std()) > 0.05
    train_imgs, test_img = imgs[:N_IMG], imgs[N_IMG:]
    student = NeRF(geo.D, geo.WD, 63, 27).to(DEV)
    optim = torch.optim.Adam(student.parameters(), lr=1e-3, betas=(0.9, 0.999))
    crit = torch.nn.MSELoss()
    getter = harness.global_batch(train_imgs, K, poses[:N_IMG], list(range(N_IMG)), (HS, WS), DEV,
                                  generator=torch.Generator(device=DEV).manual_seed(3))
    before = harness.test(0, [0], posenc, student, test_img, K, poses[N_IMG:].to(DEV), (HS, WS), opts)["psnr"][0]
    w0 = max(float(p.detach().abs().max()) for p in student.parameters())
    for it in range(1, N_STEPS + 1):
        out = harness.train(it, list(range(N_IMG)), train_imgs, (K, poses.numpy()), (HS, WS), student, crit, posenc, optim, getter, None, opts,
                            log_dir=tmp)
        if it % max(1, N_STEPS // 8) == 0:                           # the usual NeRF schedule decays the rate exponentially: 1e-3 -> 1e-4 over the run
            for g in optim.param_groups:
                g["lr"] = 1e-3 * 0.1 ** (it / N_STEPS)
    train_psnr = float(out["psnr_f" if geo.NF > 0 else "psnr_c"])
    # the trained weights travel through the reference's checkpoint format (train.py:105-114 -> test.py:20-21)
    model = NeRF(geo.D, geo.WD, 63, 27).to(DEV)
    ck = harness.load_checkpoint(harness._ckpt_path(tmp, opts.exp_name, N_STEPS), model)
    sd = {k: v.cpu().numpy() for k, v in ck["model_state_dict"].items()}
    after = harness.test(N_STEPS, [0], posenc, model, test_img, K, poses[N_IMG:].to(DEV), (HS, WS), opts)["psnr"][0]
    w1 = max(float(np.abs(v).max()) for v in sd.values())
    print(f"\n[{scene}] {N_STEPS} steps of {N_RAYS} rays: held-out PSNR {before:.2f} -> {after:.2f} dB (last training batch {train_psnr:.2f} dB); "
          f"max |weight| {w0:.3f} (init) -> {w1:.3f} (trained)")
    assert after > before + 4.0, (before, after)               # it learnt the scene: these are trained weights, not the initialisation
    return SimpleNamespace(scene=scene, geo=geo, model=model, sd=sd, K=K, poses=poses, posenc=posenc, test_img=test_img, psnr=after)


def test_trained_render_rays_vs_oracle(trained):
    """(i) 1024 full-resolution rays of a training pose (lego 800 x 800; llff: 378 x 504 through the NDC warp), 64 + 128 samples, injected
    randoms, TRAINED weights."""
    n = 1024
    geo = trained.geo
    K, H, W = geo.Kf, geo.Hf, geo.Wf
    pix = torch.from_numpy(synthetic.pixel_batch(H, W, n, 1)).to(DEV)
    o, d = ops.make_o_d_pixels(W, H, K, trained.poses[2][:3, :4].numpy(), pix)
    if geo.llff:
        o, d = ops.ndc_rays(H, W, float(K[0][0]), 1.0, o, d)                       # nerf_process.py:224-226 (the warp itself: F2)
    rays = torch.cat([o, d], -1).contiguous()
    g = torch.Generator().manual_seed(21)
    t_rand, u = torch.rand(n, SC, generator=g), (torch.rand(n, geo.NF, generator=g) if geo.NF > 0 else None)
    opts = _opts(geo)
    cfg = _cfg(geo)
    with torch.no_grad():
        got = NP.render_rays(rays, trained.model, trained.posenc, opts, t_rand=t_rand, u=u, return_intermediates=True)
    ref = R.render_rays(rays.cpu(), trained.sd, cfg, t_rand, u)
    e_c = float((got["rgb_c"].cpu() - ref["rgb_c"]).abs().max())
    e_raw_c = float((got["_raw_c"].cpu().reshape(n, SC, 4) - ref["_raw_c"]).abs().max())
    e_dc = float((got["disp_c"].cpu() - ref["disp_c"]).abs().max())
    if geo.NF == 0:                                              # coarse only: nothing is resampled, so nothing to pin
        print(f"\n[{trained.scene}] trained weights, {n} rays (coarse only, {geo.D} x {geo.WD}): rgb_c max err {e_c:.2e}, disp_c {e_dc:.2e} (raw_c {e_raw_c:.2e}, "
              f"max |raw| {float(ref['_raw_c'].abs().max()):.1f}, max sigma {float(ref['_raw_c'][..., 3].max()):.1f}); rays hitting the scene: "
              f"{float((ref['_acc_c'] > 0.5).float().mean()) * 100:.0f} %")
        assert "rgb_f" not in got and e_c <= 2e-5 and e_dc <= 2e-5, (e_c, e_dc)
        return
    packed = weights.PackedNeRF.from_state_dict(trained.sd, DEV)
    z_f = ref["_z_f"].to(DEV)
    rgb_pin, disp_pin = ops.composite(ops.mlp_rays(packed.net, packed.fine, rays, z_f), z_f, rays)[:2]
    e_f = float((rgb_pin.cpu() - ref["rgb_f"]).abs().max())
    e_disp = float((disp_pin.cpu() - ref["disp_f"]).abs().max())
    per_ray = (got["rgb_f"].cpu() - ref["rgb_f"]).abs().max(-1)[0]
    bad = float((per_ray > 1e-4).float().mean())
    acc = ref["_acc_f"]
    print(f"\n[{trained.scene}] trained weights, {n} rays: rgb_c max err {e_c:.2e} (raw_c {e_raw_c:.2e}, max |raw| {float(ref['_raw_c'].abs().max()):.1f}), "
          f"rgb_f depth-pinned {e_f:.2e} (disp {e_disp:.2e}), un-pinned rays beyond 1e-4: {bad * 100:.2f} % (worst {float(per_ray.max()):.2e}); "
          f"rays hitting the scene (acc > 0.5): {float((acc > 0.5).float().mean()) * 100:.0f} %, max sigma {float(ref['_raw_f'][..., 3].max()):.1f}")
    assert e_c <= 2e-5 and e_f <= 2e-5, (e_c, e_f)
    # un-pinned rays: sample_pdf is discontinuous, so an fp32 evaluation differs from ANY other evaluation of the same formula on about 1 % of
    # rays -- SURVEY.md section 7 measured the reference against itself (fp32 vs fp64): 9-13 of 1024 rays beyond 1e-4 on Xavier weights.  The share
    # depends on the weights (observed on these scenes over the rounds: 0.00-1.27 %); the bar is twice the reference's own typical share.
    assert bad <= 0.02, bad


def _oracle_frame(trained, seed, i_frame):
    geo = trained.geo
    o, d = R.make_o_d(geo.WS, geo.HS, trained.K, trained.poses[N_IMG][:3, :4])
    o, d = o.reshape(-1, 3), d.reshape(-1, 3)
    if geo.llff:
        o, d = R.ndc_rays(geo.HS, geo.WS, float(trained.K[0][0]), 1.0, o.expand_as(d), d)
    rays = torch.cat([o.expand_as(d), d], -1)
    s = (seed * 0x9E3779B1 + i_frame) & 0xFFFFFFFF                                 # nerf_process._next_seed
    t_rand = torch.from_numpy(R.counter_uniform(s, 0, 0, rays.shape[0], SC))
    u = torch.from_numpy(R.counter_uniform(s, 1, 0, rays.shape[0], geo.NF)) if geo.NF > 0 else None
    return R.render_rays(rays, trained.sd, _cfg(geo), t_rand, u), rays


def test_trained_heldout_frame_psnr_vs_oracle(trained):
    """(ii) the held-out pose through harness.test (test.py:38-67) vs the oracle's frame of the same pose, both against ground truth."""
    opts = _opts(trained.geo)
    HS, WS = trained.geo.HS, trained.geo.WS
    ref, _ = _oracle_frame(trained, 9, 0)
    gt = trained.test_img[0].reshape(-1, 3).cpu()
    want = float(R.mse2psnr(R.img2mse(_final(trained.geo, ref), gt)))
    NP.manual_seed(9)
    res = harness.test(N_STEPS, [0], trained.posenc, trained.model, trained.test_img, trained.K, trained.poses[N_IMG:].to(DEV), (HS, WS), opts,
                       keep_frames=True)
    got = res["psnr"][0]
    diff8 = np.abs(res["frames"][0][0].astype(np.int32) - R.to8b(_final(trained.geo, ref).reshape(HS, WS, 3).numpy()).astype(np.int32))
    print(f"\n[{trained.scene}] held-out frame: PSNR HIP {got:.4f} dB, oracle {want:.4f} dB (delta {got - want:+.4f}); 8-bit frames differ in "
          f"{(diff8 > 0).mean() * 100:.2f} % of bytes, by at most {diff8.max()}")
    assert abs(got - want) < 0.05, (got, want)
    assert diff8.max() <= 2 and (diff8 > 0).mean() < 0.02


def test_trained_reduced_precision_frames(trained):
    """(iii) the held-out frame in bf16 and f16s: PSNR against the fp32 frame of the same jitter and the change against ground truth."""
    gt = trained.test_img[0].reshape(-1, 3)
    pose = trained.poses[N_IMG].to(DEV)
    geo = trained.geo
    HS, WS = geo.HS, geo.WS
    frames = {}
    # "f16s+bf16" (MI_NERF_MODE_F16S_BF16): coarse network in split precision, fine network in bf16 -- the bf16 frame with the fp32 path's sample positions
    modes = ("fp32", "f16s", "bf16") + (("f16s+bf16",) if geo.NF > 0 else ())
    with torch.no_grad():
        for mode in modes:
            NP.manual_seed(9)
            frames[mode] = harness._render_pose(harness._frozen(trained.model, _opts(geo, precision=mode)), trained.posenc, trained.K, pose, (HS, WS),
                                                _opts(geo, precision=mode))[0]
    psnr = lambda a, b: float(-10.0 * torch.log10(torch.mean((a - b) ** 2)))
    base = psnr(frames["fp32"], gt)
    line = [f"fp32 {base:.3f} dB vs ground truth"]
    seen = {}
    for mode, floor, dmax in (("f16s", 70.0, 0.01), ("bf16", 36.0, 0.5), ("f16s+bf16", 36.0, 0.5)):   # observed: f16s 94 / 118 dB, bf16 49.5 / 47.8 dB (-0.01 / +0.13 dB)
        if mode not in frames:
            continue
        vs32, vsgt = psnr(frames[mode], frames["fp32"]), psnr(frames[mode], gt)
        worst = float((frames[mode] - frames["fp32"]).abs().max())
        seen[mode] = (vs32, worst)
        line.append(f"{mode}: {vs32:.1f} dB vs the fp32 frame (max |d rgb| {worst:.2e}), {vsgt:.3f} dB vs ground truth ({vsgt - base:+.3f})")
        assert torch.isfinite(frames[mode]).all()
        assert vs32 > floor and abs(vsgt - base) < dmax, (mode, vs32, vsgt, base)
    if "f16s+bf16" in seen:          # keeping the coarse pass fp32-grade must not make the bf16 frame worse (it removes the moved sample positions)
        assert seen["f16s+bf16"][0] > seen["bf16"][0] - 1.0, seen
    print(f"\n[{trained.scene}] held-out frame, reduced precision: " + "; ".join(line))


def test_trained_activation_range(trained):
    """(iv) the largest |pre-activation| of the trained nets over the held-out frame's points (oracle restatement of NeRF.py:33-52 run by
    torch on the device: a report, not a product path) against 65 504, the f16 range the split-precision mode computes in."""
    ref, rays = _oracle_frame(trained, 9, 0)
    sd_dev = {k: torch.as_tensor(v).to(DEV) for k, v in trained.sd.items()}
    top = {}
    for prefix, z in (("model_coarse.", ref["_z_c"]),) + ((("model_fine.", ref["_z_f"]),) if trained.geo.NF > 0 else ()):
        taps = {}
        x = ops.embed(rays.to(DEV), z.to(DEV).contiguous(), 10, 4)
        with torch.no_grad():
            raw = R.mlp_forward(sd_dev, prefix, x, trained.geo.D, 63, 27, (4,), taps=taps)
        top[prefix] = max(max(float(v.abs().max()) for v in taps.values()), float(raw.abs().max()))
    wmax = max(float(np.abs(v).max()) for v in trained.sd.values())
    print(f"\n[{trained.scene}] largest |pre-activation| coarse {top['model_coarse.']:.1f}, fine {top.get('model_fine.', float('nan')):.1f}; largest |weight| {wmax:.2f}; "
          f"f16 range 65504 -> headroom {65504.0 / max(top.values()):.0f}x")
    assert max(top.values()) < 65504.0 / 8 and wmax < 65504.0
