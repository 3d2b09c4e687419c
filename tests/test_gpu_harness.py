"""GPU parity of the callers either side of the path: device metrics, 8-bit frames,
global-batch staging, epoch cursor, per-image sampling, and the eval / video harness end to end from a checkpoint in the
reference's format.  Expected values: fixture F10 (outputs of the reference's own utils.py / rays.py) and the CPU oracle."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from nerf_pytorch_paeng_amd import harness, ops, synthetic
from nerf_pytorch_paeng_amd.model import NeRF, get_positional_encoder
from oracle import restate as R

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
T = torch.from_numpy


def test_to8b_nanmax_metrics_F10(golden):
    g = golden("F10_callers")
    assert np.array_equal(ops.to8b(T(g["to8b_in"]).to(DEV)).cpu().numpy(), g["to8b_out"])                 # bit-exact bytes
    disp = T(g["disp_in"]).to(DEV)
    mx = ops.nanmax(disp)
    assert float(mx) == float(g["disp_max"])
    assert np.array_equal(ops.to8b(disp, mx).cpu().numpy(), g["disp8"])
    assert float(ops.nanmax(T(g["nanmax_in"]).to(DEV))) == float(g["nanmax_out"])
    assert torch.isnan(ops.nanmax(torch.full((5,), float("nan"), device=DEV))).all()
    m = ops.image_metrics(T(g["metric_pred"]).to(DEV), T(g["metric_target"]).to(DEV)).cpu()
    assert abs(float(m[0]) - float(g["metric_mse"])) <= 1e-6 * float(g["metric_mse"])                     # summation order only
    assert abs(float(m[1]) - float(g["metric_psnr"][0])) < 1e-4
    # a frame-sized reduction against float64
    rs = np.random.RandomState(3)
    a, b = rs.uniform(0, 1, (800 * 800, 3)).astype(np.float32), rs.uniform(0, 1, (800 * 800, 3)).astype(np.float32)
    want = float(np.mean((a.astype(np.float64) - b) ** 2))
    got = ops.image_metrics(T(a).to(DEV), T(b).to(DEV)).cpu()
    assert abs(float(got[0]) - want) < 2e-7 * want and abs(float(got[1]) + 10 * np.log10(want)) < 1e-4
    with pytest.raises(ops.MiNerfError):
        ops.image_metrics(torch.zeros(4, 3, device=DEV), torch.zeros(5, 3, device=DEV))


def test_global_batch_F10_and_oracle(golden):
    g = golden("F10_callers")
    H, W = (int(v) for v in g["gb_HW"])
    it = list(g["gb_i_train"])
    getter = harness.global_batch(g["gb_images"], g["gb_K"], g["gb_poses"], it, (H, W), DEV, shuffle=False)
    # get_rays_np computes in the dtype numpy promotes to (float64 here) and main.py:101 rounds to fp32 at the end; the
    # kernel computes in fp32 throughout -> 1-ulp differences in the directions, origins and pixels exact
    got = getter.rays_rgb.cpu().numpy()
    np.testing.assert_array_equal(got[:, 0], g["gb_rays_rgb"][:, 0])
    np.testing.assert_array_equal(got[:, 2], g["gb_rays_rgb"][:, 2])
    np.testing.assert_allclose(got[:, 1], g["gb_rays_rgb"][:, 1], rtol=3e-7, atol=1e-7)
    # lego-sized cameras, more images, against the oracle
    K, Hh, Ww = synthetic.lego_camera()
    Hs, Ws = 40, 50
    Ks = np.array([[K[0][0] * Ws / Ww, 0, Ws / 2], [0, K[1][1] * Hs / Hh, Hs / 2], [0, 0, 1]])
    poses = np.stack([synthetic.pose_spherical(a, -30.0, 4.0) for a in np.linspace(-180, 180, 6)[:-1]], 0).astype(np.float32)
    imgs = np.random.RandomState(5).uniform(0, 1, (5, Hs, Ws, 3)).astype(np.float32)
    want = R.rays_rgb_global_batch(Hs, Ws, Ks, poses, imgs, [4, 0, 2])
    got = harness.global_batch(imgs, Ks, poses, [4, 0, 2], (Hs, Ws), DEV, shuffle=False).rays_rgb.cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=3e-7, atol=1e-7)
    # shuffled: same multiset of rows (np.random.shuffle permutes the leading axis, main.py:102)
    gen = torch.Generator(device=DEV).manual_seed(1)
    shuffled = harness.global_batch(imgs, Ks, poses, [4, 0, 2], (Hs, Ws), DEV, generator=gen).rays_rgb
    assert isinstance(shuffled, harness.ShuffledRows) and shuffled.shape == (3 * Hs * Ws, 3, 3)      # a permutation beside the table, no second table
    sh = shuffled.materialize().cpu().numpy()
    assert not np.array_equal(sh, got)
    for a, b in ((0, 7), (5, 5 + 300), (3 * Hs * Ws - 11, 3 * Hs * Ws)):                             # the slices a training step takes (train.py:29)
        np.testing.assert_array_equal(shuffled[a:b].cpu().numpy(), sh[a:b])
    with pytest.raises(ops.MiNerfError):
        shuffled[3]
    key = lambda a: a.reshape(a.shape[0], -1)[np.lexsort(a.reshape(a.shape[0], -1).T[::-1])]
    np.testing.assert_array_equal(key(sh), key(got))


def test_permute_rows_and_epoch_cursor(golden):
    g = golden("F10_callers")
    src = torch.arange(7 * 9, dtype=torch.float32, device=DEV).reshape(7, 3, 3)
    perm = torch.tensor([3, 0, 6, 5, 1, 2, 4], device=DEV)
    assert torch.equal(ops.permute_rows(src, perm), src[perm])
    with pytest.raises(ops.MiNerfError):
        ops.permute_rows(src, perm[:5])
    # the gather behind it, at sizes that cross block (256 rows) and grid-stride boundaries, rows of 9 floats (the [3][3] rows) and of other widths
    gen = torch.Generator(device=DEV).manual_seed(4)
    for n_src, n_out, shape in ((1000, 1000, (3, 3)), (5000, 257, (3, 3)), (300, 4096, (3, 3)), (70000, 70000, (3, 3)), (513, 513, (5,)), (64, 1, (32,)),
                                (256 * 65536 + 300, 256 * 65536 + 300, (1,))):
        table = torch.rand(n_src, *shape, device=DEV)
        idx = torch.randint(0, n_src, (n_out,), device=DEV, generator=gen) if n_out != n_src else torch.randperm(n_src, device=DEV, generator=gen)
        assert torch.equal(ops.gather_rows(table, idx), table[idx]), (n_src, n_out, shape)
    assert ops.gather_rows(src, perm[:0]).shape == (0, 3, 3)
    with pytest.raises(ops.MiNerfError):
        ops.gather_rows(src, perm.int())
    # (i_batch, epoch) trace of utils.GetterRayBatchIdx: 10 rows, batch 4, 7 calls
    table = torch.arange(30, dtype=torch.float32, device=DEV).reshape(10, 3)
    getter = harness.GetterRayBatchIdx(table)
    epochs_seen = set()
    for want in g["cursor_trace"]:
        i_batch, rr, epoch = getter(4)
        assert (i_batch, epoch) == (int(want[0]), int(want[1]))
        whole = rr if isinstance(rr, torch.Tensor) else rr.materialize()
        assert torch.equal(torch.sort(whole[:, 0]).values.cpu(), torch.arange(0, 30, 3, dtype=torch.float32))   # still a permutation
        batch = rr[i_batch - 4:i_batch]                                                                   # train.py:29
        assert batch.shape == (4, 3) and torch.equal(batch, whole[i_batch - 4:i_batch])
        assert (epoch == 0) == isinstance(rr, torch.Tensor)              # unshuffled until the first epoch ends, a ShuffledRows view afterwards
        epochs_seen.add(epoch)
    assert getter.table is table and len(epochs_seen) > 1               # the table itself never moved


def test_sample_rays_and_pixel():
    K, H, W = synthetic.lego_camera()
    H, W = 64, 48
    K = np.array([[60.0, 0, W / 2], [0, 60.0, H / 2], [0, 0, 1]])
    pose = torch.from_numpy(synthetic.pose_spherical(20.0, -30.0, 4.0)).float().to(DEV)
    img = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(0)).to(DEV)
    o_full, d_full = harness.make_o_d(W, H, K, pose[:3, :4])
    for it, opts in ((0, SimpleNamespace(N_rays=200, precrop_iters=500, precrop_frac=0.5)),
                     (900, SimpleNamespace(N_rays=1024, precrop_iters=500, precrop_frac=0.5))):
        ro, rd, tgt = harness.sample_rays_and_pixel(it, W, H, K, pose[:3, :4], img, opts, generator=torch.Generator(device=DEV).manual_seed(2))
        assert ro.shape == rd.shape == tgt.shape == (opts.N_rays, 3)
        # recover the pixel of every sample from its target colour (continuous random image: unique colours)
        flat = img.reshape(-1, 3)
        idx = torch.cdist(tgt, flat).argmin(1)
        assert idx.unique().numel() == opts.N_rays                                          # without replacement (rays.py:53-54)
        assert torch.equal(flat[idx], tgt)
        assert torch.allclose(rd, d_full.reshape(-1, 3)[idx], rtol=0, atol=0) and torch.equal(ro, o_full.reshape(-1, 3)[idx])
        py, px = idx // W, idx % W
        if it < opts.precrop_iters:                                                         # centre crop (rays.py:39-44)
            dH, dW = int(H // 2 * 0.5), int(W // 2 * 0.5)
            assert int(py.min()) >= H // 2 - dH and int(py.max()) <= H // 2 + dH - 1
            assert int(px.min()) >= W // 2 - dW and int(px.max()) <= W // 2 + dW - 1
    with pytest.raises(ops.MiNerfError):
        harness.sample_rays_and_pixel(900, W, H, K, pose[:3, :4], img, SimpleNamespace(N_rays=H * W + 1, precrop_iters=0, precrop_frac=0.5))


def test_eval_and_video_harness_from_reference_checkpoint(tmp_path):
    """test.py:17-108 / 111-174 end to end: checkpoint in the reference's format -> frames, PSNR, _result.txt.  The oracle
    renders the same poses on the CPU; dataset-level bar of the north star: PSNR within 0.05 dB."""
    D, Wd, Hs, Ws = 4, 128, 20, 24
    sd = synthetic.make_state_dict(13, D, Wd)
    src = NeRF(D, Wd, 63, 27)
    src.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    exp, idx = "lego_t", 2000
    os.makedirs(tmp_path / exp)
    torch.save({"idx": idx, "model_state_dict": src.state_dict(), "optimizer_state_dict": {}}, harness._ckpt_path(str(tmp_path), exp, idx))
    model = NeRF(D, Wd, 63, 27).to(DEV)                                   # fresh weights: the harness must load the checkpoint
    posenc = get_positional_encoder(10), get_positional_encoder(4)
    K = np.array([[30.0, 0, Ws / 2], [0, 30.0, Hs / 2], [0, 0, 1]])
    poses = harness.get_render_pose(n_angle=3, phi=-30.0, nf=4.0)
    opts = SimpleNamespace(near=2.0, far=6.0, N_samples_c=32, N_samples_f=32, perturb=0.0, chunk_rays=4096, chunk_pts=524288,
                           data_type="blender", gpu_ids=[0], rank=0, exp_name=exp, n_angle=3, single_angle=-1, phi=-30.0, nf=4.0)
    cfg = R.PathConfig(near=2.0, far=6.0, N_samples_c=32, N_samples_f=32, perturb=0.0, netDepth=D, netWidth=Wd)
    gt = torch.rand(3, Hs, Ws, 3, generator=torch.Generator().manual_seed(4))
    # oracle frames.  perturb = 0 makes the RESAMPLING deterministic; the coarse depths are jittered regardless
    # (nerf_process.py:58-60 draws unconditionally).  The product's jitter is counter-based: frame i of a harness run
    # started at manual_seed(7) uses seed (7 * 0x9E3779B1 + i) mod 2^32, which the numpy mirror reproduces for the oracle.
    from nerf_pytorch_paeng_amd import nerf_process as NP
    want_rgb, want_psnr = [], []
    for i in range(3):
        o, d = R.make_o_d(Ws, Hs, K, poses[i][:3, :4])
        rays = torch.cat([o.reshape(-1, 3), d.reshape(-1, 3)], -1)
        t_rand = torch.from_numpy(R.counter_uniform((7 * 0x9E3779B1 + i) & 0xFFFFFFFF, 0, 0, rays.shape[0], 32))
        out = R.render_rays(rays, sd, cfg, t_rand)
        want_rgb.append(out["rgb_f"].reshape(Hs, Ws, 3))
        want_psnr.append(float(R.mse2psnr(R.img2mse(out["rgb_f"], gt[i].reshape(-1, 3)))))
    save_dir = str(tmp_path / "test_result")
    NP.manual_seed(7)
    res = harness.test(idx, [0, 1, 2], posenc, model, gt.to(DEV), K, poses.to(DEV), (Hs, Ws), opts, log_dir=str(tmp_path), save_dir=save_dir,
                       keep_frames=True)
    assert len(res["psnr"]) == 3 and res["ssim"] is None
    for i in range(3):
        assert abs(res["psnr"][i] - want_psnr[i]) < 0.05, (i, res["psnr"][i], want_psnr[i])
        rgb8 = res["frames"][i][0]
        diff = np.abs(rgb8.astype(np.int32) - R.to8b(want_rgb[i].numpy()).astype(np.int32))
        assert diff.max() <= 1 and (diff > 0).mean() < 0.02                                 # truncation flips at byte boundaries only
        assert os.path.exists(os.path.join(save_dir, f"{i:03d}.png")) and os.path.exists(os.path.join(save_dir, f"{i:03d}_disp.png"))
    txt = open(os.path.join(save_dir, "_result.txt")).read()
    assert txt.count("idx:") == 3 and "Mean Value ) PSNR" in txt and abs(res["mean_psnr"] - np.mean(want_psnr)) < 0.05
    # video path: poses regenerated from opts for blender data (test.py:119-124), same frames as the eval path
    NP.manual_seed(7)
    rgbs, disps = harness.render(idx, posenc, model, K, None, (Hs, Ws), opts, log_dir=str(tmp_path))
    assert rgbs.shape == (3, Hs, Ws, 3) and disps.shape == (3, Hs, Ws) and rgbs.dtype == np.uint8
    for i in range(3):
        assert np.array_equal(rgbs[i], res["frames"][i][0]) and np.array_equal(disps[i], res["frames"][i][1][:, :, 0])


def test_train_loop_global_batch_and_per_image_with_checkpoint(tmp_path):
    """train.py:12-119 through harness.train: both ray-selection modes, checkpoint in the reference's format, reload by
    harness.test; the loss on a fixed tiny scene must come down."""
    D, Wd, Hs, Ws, n_img = 4, 128, 16, 16, 3
    torch.manual_seed(0)                                                # Xavier init of the model (NeRF.py:63-65) is the only unseeded draw
    np.random.seed(0)                                                   # per-image mode picks the image with np.random.choice (train.py:37)
    model = NeRF(D, Wd, 63, 27).to(DEV)
    posenc = get_positional_encoder(10), get_positional_encoder(4)
    K = np.array([[24.0, 0, Ws / 2], [0, 24.0, Hs / 2], [0, 0, 1]])
    poses = harness.get_render_pose(n_angle=n_img, phi=-30.0, nf=4.0).numpy()
    images = np.random.RandomState(1).uniform(0.2, 0.8, (n_img, Hs, Ws, 3)).astype(np.float32)
    opts = SimpleNamespace(near=2.0, far=6.0, N_samples_c=16, N_samples_f=16, perturb=1.0, chunk_rays=4096, chunk_pts=524288,
                           data_type="blender", gpu_ids=[0], rank=0, exp_name="tiny", N_rays=128, global_batch=True, idx_save=4,
                           precrop_iters=2, precrop_frac=0.5)
    optim = torch.optim.Adam(model.parameters(), lr=5e-3, betas=(0.9, 0.999))
    crit = torch.nn.MSELoss()
    getter = harness.global_batch(images, K, poses, [0, 1, 2], (Hs, Ws), DEV, generator=torch.Generator(device=DEV).manual_seed(0))
    losses = []
    for it in range(1, 9):
        out = harness.train(it, [0, 1, 2], images, (K, poses), (Hs, Ws), model, crit, posenc, optim, getter, None, opts, log_dir=str(tmp_path))
        losses.append(float(out["loss"]))
    assert np.mean(losses[-3:]) < np.mean(losses[:3]), losses
    assert getter.epoch >= 1                                            # 768 rays, 128 per step: the cursor wrapped and reshuffled
    ck = torch.load(harness._ckpt_path(str(tmp_path), "tiny", 8), map_location="cpu", weights_only=False)
    assert ck["idx"] == 8 and set(ck) == {"idx", "model_state_dict", "optimizer_state_dict"}
    assert os.path.exists(harness._ckpt_path(str(tmp_path), "tiny", 4))
    # per-image mode (global_batch off), with the early centre crop
    opts.global_batch = False
    opts.N_rays = 32
    for it in range(1, 4):
        out = harness.train(it, [0, 1, 2], images, (K, poses), (Hs, Ws), model, crit, posenc, optim, None, None, opts,
                            generator=torch.Generator(device=DEV).manual_seed(it))
        assert torch.isfinite(out["loss"]) and "psnr_f" in out
    # evaluate the step-8 checkpoint with the eval harness
    fresh = NeRF(D, Wd, 63, 27).to(DEV)
    res = harness.test(8, [0], posenc, fresh, torch.from_numpy(images[:1]).to(DEV), K, torch.from_numpy(poses[:1]).to(DEV), (Hs, Ws), opts,
                       log_dir=str(tmp_path))
    assert len(res["psnr"]) == 1 and np.isfinite(res["psnr"][0])
    for (k, a), (_, b) in zip(fresh.state_dict().items(), ck["model_state_dict"].items()):
        assert torch.equal(a.cpu(), b), k


@pytest.mark.parametrize("Wd,precision", [(128, "fp32"), (256, "fp32"), (256, "f16s")])
def test_student_learns_a_teacher_scene(Wd, precision):
    """A functional check of the whole training path: images of a fixed random 'teacher' NeRF (rendered by the inference
    kernels) are the ground truth; a freshly initialised student trained with harness.train on the global batch must approach
    them -- PSNR on a held-out pose rises by more than 6 dB in 600 Adam steps of 512 rays (seeded: 9.7 -> ~19 dB).
    ``precision`` "f16s": the same through the split-precision step (opts.precision; eval frames in split precision too)."""
    from nerf_pytorch_paeng_amd import nerf_process as NP
    torch.manual_seed(1)
    NP.manual_seed(0)
    D, Hs, Ws, n_img = 4, 24, 24, 6
    posenc = get_positional_encoder(10), get_positional_encoder(4)
    K = np.array([[36.0, 0, Ws / 2], [0, 36.0, Hs / 2], [0, 0, 1]])
    poses = harness.get_render_pose(n_angle=n_img + 1, phi=-30.0, nf=4.0)
    opts = SimpleNamespace(near=2.0, far=6.0, N_samples_c=24, N_samples_f=24, perturb=1.0, chunk_rays=4096, chunk_pts=524288,
                           data_type="blender", gpu_ids=[0], rank=0, exp_name="teach", N_rays=512, global_batch=True, idx_save=0,
                           n_angle=n_img + 1, single_angle=-1, phi=-30.0, nf=4.0, precision=precision)
    teacher = NeRF(D, Wd, 63, 27).to(DEV)
    teacher.load_state_dict({k: torch.as_tensor(v) for k, v in synthetic.make_state_dict(77, D, Wd).items()})
    with torch.no_grad():
        imgs = torch.stack([harness._render_pose(teacher, posenc, K, poses[i].to(DEV), (Hs, Ws), opts)[0].reshape(Hs, Ws, 3) for i in range(n_img + 1)], 0)
    train_imgs, test_img = imgs[:n_img], imgs[n_img:]
    assert float(train_imgs.std()) > 0.05                                # the teacher scene is not flat
    student = NeRF(D, Wd, 63, 27).to(DEV)
    optim = torch.optim.Adam(student.parameters(), lr=2e-3, betas=(0.9, 0.999))
    crit = torch.nn.MSELoss()
    getter = harness.global_batch(train_imgs, K, poses[:n_img], list(range(n_img)), (Hs, Ws), DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    before = harness.test(0, [0], posenc, student, test_img, K, poses[n_img:].to(DEV), (Hs, Ws), opts)["psnr"][0]
    for it in range(1, 601):
        harness.train(it, list(range(n_img)), train_imgs, (K, poses.numpy()), (Hs, Ws), student, crit, posenc, optim, getter, None, opts)
    after = harness.test(600, [0], posenc, student, test_img, K, poses[n_img:].to(DEV), (Hs, Ws), opts)["psnr"][0]
    print(f"W={Wd} {precision}: held-out PSNR {before:.1f} dB -> {after:.1f} dB")
    assert after > before + 6.0, (before, after)


def test_example_script_runs_the_references_main_loop(tmp_path):
    """examples/train_eval_render.py: train -> checkpoint -> test -> render (main.py:136-158) on a tiny scene, at a width without kernels of its own."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("train_eval_render", os.path.join(root, "examples", "train_eval_render.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.main(["--steps", "60", "--size", "16", "--views", "4", "--out", str(tmp_path), "--net-width", "64", "--render-views", "2"])
    assert len(res["psnr"]) == 2 and all(np.isfinite(res["psnr"]))
    assert os.path.exists(tmp_path / "test_result" / "_result.txt") and os.path.exists(tmp_path / "render_result" / "1_rgb.png")
    assert os.path.exists(harness._ckpt_path(str(tmp_path), "demo", 60))
