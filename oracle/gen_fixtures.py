#!/usr/bin/env python3
"""Generate the golden fixtures F1..F12 (F10/F12: the callers either side of the path; F11: the reference's own
training-step gradients) by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference).  It imports the reference's hot-path
modules unmodified, drives them on CPU through a small import shim, and writes inputs + the
reference's outputs as compressed ``.npz`` files under ``tests/golden/``.  No reference source
text is copied anywhere; fixtures hold data only.  Weights are never stored: both sides
regenerate them from ``nerf_pytorch_paeng_amd.synthetic.make_state_dict(seed, ...)``.

Shim (container only):
  * ``IQA_pytorch`` and ``cv2`` are imported-but-unused by the path (utils.py:3, model/NeRF.py:5)
    and absent here -> empty stub modules.
  * the module-global name ``torch`` inside ``nerf_process`` is replaced by a proxy that forwards
    to real torch except ``device(...)`` -> cpu (the reference hard-codes ``cuda:N``,
    nerf_process.py:45-59,158-163) and ``rand(...)`` -> replay of injected tensors (so t_rand / u
    are known; draw order inside one render_rays call is t_rand then u).
  * ``Tensor.get_device`` -> 'cpu' (nerf_process.py:94, rays.py:23-24).

Usage:  python oracle/gen_fixtures.py            (rewrites tests/golden/*.npz)
        python oracle/gen_fixtures.py --f10 | --f11 | --f12   (one of the later fixtures only)
"""
from __future__ import annotations

import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("NERF_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from nerf_pytorch_paeng_amd import synthetic  # noqa: E402
from oracle.restate import counter_uniform  # noqa: E402


# ----------------------------------------------------------------------------------------------
# import shim
# ----------------------------------------------------------------------------------------------
class _TorchProxy:
    """Forwards to torch; pins devices to CPU and replays injected uniforms."""

    def __init__(self):
        self.queue = []          # tensors to hand out from rand(), FIFO
        self.drawn = []          # shapes requested (for the draw-order record)

    def __getattr__(self, name):
        return getattr(torch, name)

    def device(self, *a, **k):
        return torch.device("cpu")

    def rand(self, *size, **kw):
        shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (list, tuple, torch.Size)) else tuple(size)
        self.drawn.append(shape)
        if not self.queue:
            raise RuntimeError(f"reference asked for rand{shape} but nothing was injected")
        t = self.queue.pop(0)
        assert tuple(t.shape) == shape, (tuple(t.shape), shape)
        return t.clone()


def load_reference():
    for name in ("IQA_pytorch", "cv2"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            if name == "IQA_pytorch":
                m.SSIM = m.LPIPSvgg = object
            sys.modules[name] = m
    sys.path.insert(0, REF)
    torch.Tensor.get_device = lambda self: "cpu"
    import nerf_process as ref_np
    import rays as ref_rays
    from model.NeRF import NeRF as RefNeRF
    from model.PositionalEncoding import get_positional_encoder as ref_posenc
    proxy = _TorchProxy()
    ref_np.torch = proxy
    return SimpleNamespace(np=ref_np, rays=ref_rays, NeRF=RefNeRF, posenc=ref_posenc, proxy=proxy)


def make_opts(**kw):
    base = dict(near=2.0, far=6.0, N_samples_c=64, N_samples_f=128, perturb=1.0, chunk_rays=4096,
                chunk_pts=524288, data_type="blender", gpu_ids=[0], rank=0)
    base.update(kw)
    return SimpleNamespace(**base)


def ref_model(R, seed, D, W, L_x=10, L_d=4):
    in_x, in_d = 3 + 6 * L_x, 3 + 6 * L_d
    m = R.NeRF(D, W, in_x, in_d, skips=[4], gt_camera_param=(None, None))
    sd = synthetic.make_state_dict(seed, D, W, in_x, in_d)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.eval()
    return m


def npy(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: npy(v) for k, v in arrays.items()})
    print(f"  wrote {os.path.relpath(path, REPO)}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# ----------------------------------------------------------------------------------------------
def f10_callers():
    """F10: the callers either side of the path -- utils.py metrics / 8-bit conversion /
    epoch cursor, dataset/render_pose.py camera path, rays.get_rays_np feeding the global-batch layout of main.py:92-101."""
    print("F10 callers")
    import utils as ref_utils                                   # noqa: E402  (IQA_pytorch is stubbed by load_reference)
    # dataset/__init__.py eagerly imports the disk loaders (imageio, cv2, configargparse ...): load render_pose.py by path
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_render_pose", os.path.join(REF, "dataset", "render_pose.py"))
    rp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rp)
    ref_poses, ref_pose_spherical = rp.get_render_pose, rp.pose_spherical
    import rays as ref_rays
    rs = np.random.RandomState(77)
    f = {}
    f["poses8"] = ref_poses(n_angle=8, single_angle=-1, phi=-30.0, nf=4.0)
    f["pose_single"] = ref_poses(n_angle=1, single_angle=120, phi=-20.0, nf=3.5)
    f["pose_sph"] = ref_pose_spherical(33.0, -41.0, 4.0)
    x = np.concatenate([rs.uniform(-0.3, 1.3, 500), [0.0, 1.0, 1.0 / 255, 254.999 / 255, 0.5, -0.0]]).astype(np.float32)
    f["to8b_in"], f["to8b_out"] = x, ref_utils.to8b(x)
    disp = rs.uniform(0.0, 5.0, (7, 9)).astype(np.float32)
    f["disp_in"], f["disp_max"], f["disp8"] = disp, np.nanmax(disp), ref_utils.to8b(disp / np.nanmax(disp))   # test.py:56
    with_nan = disp.copy(); with_nan[2, 3] = np.nan; with_nan[6, 8] = np.nan
    f["nanmax_in"], f["nanmax_out"] = with_nan, np.nanmax(with_nan)
    pred = torch.from_numpy(rs.uniform(0, 1, (300, 3)).astype(np.float32))
    tgt = torch.from_numpy(rs.uniform(0, 1, (300, 3)).astype(np.float32))
    mse = ref_utils.img2mse(pred, tgt)
    f["metric_pred"], f["metric_target"], f["metric_mse"], f["metric_psnr"] = pred, tgt, mse, ref_utils.mse2psnr(mse)
    # global batch: main.py:95-101 executed on the reference's get_rays_np (the statements are inline in main_worker)
    H, W, n_img = 6, 5, 3
    K = np.array([[7.5, 0, 2.5], [0, 7.5, 3.0], [0, 0, 1]])
    poses = np.stack([synthetic.pose_spherical(a, -30.0, 4.0) for a in (0.0, 77.0, -130.0)], 0).astype(np.float32)
    images = rs.uniform(0, 1, (n_img, H, W, 3)).astype(np.float32)
    i_train = [0, 2]
    rays = np.stack([ref_rays.get_rays_np(H, W, K, p) for p in poses[:, :3, :4]], 0)
    rays_rgb = np.concatenate([rays, images[:, None]], 1)
    rays_rgb = np.transpose(rays_rgb, [0, 2, 3, 1, 4])
    rays_rgb = np.stack([rays_rgb[i] for i in i_train], 0)
    rays_rgb = np.reshape(rays_rgb, [-1, 3, 3]).astype(np.float32)
    f.update(gb_K=K, gb_HW=np.array([H, W]), gb_poses=poses, gb_images=images, gb_i_train=np.array(i_train), gb_rays_rgb=rays_rgb)
    # epoch cursor: (i_batch, epoch) trace of utils.GetterRayBatchIdx for 10 rows, batch 4, 7 calls
    getter = ref_utils.GetterRayBatchIdx(torch.arange(30, dtype=torch.float32).reshape(10, 3))
    trace = []
    for _ in range(7):
        i_batch, rr, epoch = getter(4)
        trace.append([i_batch, epoch, int(torch.sort(rr[:, 0]).values.equal(torch.arange(0, 30, 3, dtype=torch.float32)))])
    f["cursor_trace"] = np.array(trace)
    save("F10_callers", **f)


def f11_train_step(R):
    """F11: the reference's own training step (train.py:53-70) on the F8 `legoA` inputs: batchify_rays_and_render_by_chunk with
    gradients enabled, criterion = MSELoss (main.py:76) on the coarse and fine colours, loss = loss_c + loss_f, loss.backward().
    The 48 parameter gradients of the 8x256 coarse/fine pair are the fixture (inputs are regenerated from seeds; the depths the
    reference sampled are F8's legoA_z_c / legoA_z_f)."""
    print("F11 training-step gradients")
    enc_x, _ = R.posenc(10)
    enc_d, _ = R.posenc(4)
    Kl, Hl, Wl = synthetic.lego_camera()
    with torch.no_grad():
        o, d = R.rays.make_o_d(Wl, Hl, Kl, torch.from_numpy(synthetic.pose_spherical(0.0, -30.0, 4.0)[:3, :4]))
    pix = synthetic.pixel_batch(Hl, Wl, 4096, 0)[:64]
    ro, rd = o.reshape(-1, 3)[pix].contiguous(), d.reshape(-1, 3)[pix].contiguous()
    f = {}
    for tag, (D, W, seed, Sc, Nf) in (("d8w256", (8, 256, 0, 64, 128)), ("d4w128", (4, 128, 2, 24, 40))):
        opts = make_opts(N_samples_c=Sc, N_samples_f=Nf)
        m = ref_model(R, seed, D, W)
        m.train()
        t_rand = torch.from_numpy(counter_uniform(0, 0, 0, 64, Sc))
        u = torch.from_numpy(counter_uniform(0, 1, 0, 64, Nf))
        target = torch.from_numpy(np.random.RandomState(11).uniform(0, 1, (64, 3)).astype(np.float32))
        criterion = torch.nn.MSELoss()                                         # main.py:76
        with torch.enable_grad():
            R.proxy.queue = [t_rand, u]
            rgb_c, disp_c, rgb_f, disp_f = R.np.batchify_rays_and_render_by_chunk(ro, rd, m, (enc_x, enc_d), Hl, Wl, Kl, opts)   # train.py:53
            loss_c = criterion(rgb_c, target)                                  # train.py:60
            loss_f = criterion(rgb_f, target)                                  # train.py:64
            loss = loss_c + loss_f                                             # train.py:66
            loss.backward()                                                    # train.py:69
        f.update({f"{tag}_target": target, f"{tag}_loss_c": loss_c.detach(), f"{tag}_loss_f": loss_f.detach(),
                  f"{tag}_rgb_c": rgb_c.detach(), f"{tag}_rgb_f": rgb_f.detach(),
                  f"{tag}_cfg": np.array([D, W, seed, Sc, Nf])})
        # the depths the reference used (staged replay, no grad), so that a checker can pin them
        with torch.no_grad():
            R.proxy.queue = [t_rand]
            emb_c, z_c, rdd = R.np.pre_process(torch.cat([ro, rd], -1), (enc_x, enc_d), opts, isFine=False)
            _, _, _, w_c, _ = R.np.post_process(m(emb_c).reshape(64, Sc, 4), z_c, rdd)
            R.proxy.queue = [u]
            _, z_f, _ = R.np.pre_process(torch.cat([ro, rd], -1), (enc_x, enc_d), opts, z_vals=z_c, weights=w_c, isFine=True)
        f[f"{tag}_z_c"], f[f"{tag}_z_f"] = z_c, z_f
        n = 0
        for k, p in m.named_parameters():
            f[f"{tag}_grad.{k}"] = p.grad.detach().clone()
            n += 1
        print(f"  {tag}: loss_c {float(loss_c):.6f} loss_f {float(loss_f):.6f}, {n} gradient tensors")
    save("F11_train_grads", **f)


def f12_llff_spiral():
    """F12: the LLFF camera path (dataset/load_llff.py:151-189 normalize / viewmatrix / poses_avg / render_path_spiral, consumed at
    :294-328 and rendered by test.py:132-145).  load_llff.py is loaded by path under a stub package (its package __init__ drags in the
    blender / custom loaders; `imageio` and `.colmap` are absent and unused here); `_load_data` (disk) is replaced by synthetic
    poses_bounds so that the reference's own `load_llff()` runs end to end."""
    print("F12 LLFF spiral")
    import importlib.util
    for name in ("imageio",):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    pkg = types.ModuleType("ref_dataset"); pkg.__path__ = [os.path.join(REF, "dataset")]
    sys.modules["ref_dataset"] = pkg
    col = types.ModuleType("ref_dataset.colmap"); col.gen_poses = lambda *a, **k: None
    sys.modules["ref_dataset.colmap"] = col
    spec = importlib.util.spec_from_file_location("ref_dataset.load_llff", os.path.join(REF, "dataset", "load_llff.py"))
    ll = importlib.util.module_from_spec(spec)
    sys.modules["ref_dataset.load_llff"] = ll
    spec.loader.exec_module(ll)

    rs = np.random.RandomState(2024)
    N, H, W, focal = 11, 6, 8, 9.5
    # forward-facing rig in the raw poses_bounds convention [3, 5, N] (columns: down, right, back, position, hwf)
    raw = np.zeros((3, 5, N))
    for i in range(N):
        a, b, c = rs.normal(0, 0.08, 3)
        Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
        Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
        Rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
        raw[:, :3, i] = Rz @ Ry @ Rx
        raw[:, 3, i] = rs.uniform(-1.5, 1.5, 3) * np.array([1.0, 0.6, 0.15])
        raw[:, 4, i] = [H, W, focal]
    bds_raw = np.stack([rs.uniform(1.1, 1.6, N), rs.uniform(7.0, 12.0, N)], 0)          # [2, N]
    imgs = rs.uniform(0, 1, (H, W, 3, N))
    f = dict(raw_poses=raw, raw_bds=bds_raw, hw=np.array([H, W]), focal=focal)
    # path_zflat=True is not captured: the reference does `N_views /= 2` (load_llff.py:321) and hands the float to np.linspace,
    # which raises TypeError under this numpy; main.py:43-44 never sets it
    for tag, zflat in (("spiral", False),):
        ll._load_data = lambda basedir, factor=None, colmap_relaunch=False: (raw.copy(), bds_raw.copy(), imgs.copy())
        images, (K, ext), hw, (i_train, i_val, i_test), render_poses = ll.load_llff("unused", downsample=8, path_zflat=zflat)
        f.update({f"{tag}_render_poses": render_poses, f"{tag}_K": K, f"{tag}_extrinsic": ext, f"{tag}_i_test": i_test, f"{tag}_i_train": i_train})
    # the building blocks on their own (load_llff.py:151-189)
    poses = np.concatenate([raw[:, 1:2, :], -raw[:, 0:1, :], raw[:, 2:, :]], 1)
    poses = np.moveaxis(poses, -1, 0).astype(np.float32)
    bds = np.moveaxis(bds_raw, -1, 0).astype(np.float32)
    sc = 1. / (bds.min() * .75)
    poses[:, :3, 3] *= sc
    bds *= sc
    rec = ll.recenter_poses(poses)
    c2w = ll.poses_avg(rec)
    v = rs.normal(0, 1, 3)
    f.update(rec_poses=rec, rec_bds=bds, poses_avg=c2w, normalize_in=v, normalize_out=ll.normalize(v),
             viewmatrix_out=ll.viewmatrix(rec[0, :3, 2], rec[1, :3, 1], rec[2, :3, 3]),
             spiral_direct=np.array(ll.render_path_spiral(c2w, ll.normalize(rec[:, :3, 1].sum(0)), np.array([0.3, 0.2, 0.1]), 3.5, 0.2, zrate=.5, rots=2, N=9)))
    save("F12_llff_spiral", **f)


def main():
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    R = load_reference()
    if "--f10" in sys.argv:                                     # only the callers fixture (F1-F9 untouched)
        f10_callers()
        return
    if "--f11" in sys.argv:
        f11_train_step(R)
        return
    if "--f12" in sys.argv:
        f12_llff_spiral()
        return
    rs = np.random.RandomState(1234)

    # ---- F1 ray generation --------------------------------------------------------------------
    print("F1 ray-gen")
    Kl, Hl, Wl = synthetic.lego_camera()
    Kf, Hf, Wf = synthetic.fern_camera()
    f1 = {}
    for tag, (K, H, W), poses in (
            ("lego", (Kl, Hl, Wl), [synthetic.pose_spherical(0.0, -30.0, 4.0), synthetic.pose_spherical(117.0, -30.0, 4.0)]),
            ("fern", (Kf, Hf, Wf), [synthetic.fern_pose(), np.eye(4, dtype=np.float32)])):
        for pi, pose in enumerate(poses):
            o, d = R.rays.make_o_d(W, H, K, torch.from_numpy(pose[:3, :4]))
            on, dn = R.rays.get_rays_np(H, W, K, pose[:3, :4])
            ys = np.concatenate([[0, 0, H - 1, H - 1], rs.randint(0, H, 60)])
            xs = np.concatenate([[0, W - 1, 0, W - 1], rs.randint(0, W, 60)])
            key = f"{tag}{pi}"
            f1.update({f"{key}_K": K, f"{key}_HW": np.array([H, W]), f"{key}_pose": pose,
                       f"{key}_ys": ys, f"{key}_xs": xs,
                       f"{key}_o": npy(o)[ys, xs], f"{key}_d": npy(d)[ys, xs],
                       f"{key}_np_o": on[ys, xs], f"{key}_np_d": dn[ys, xs]})
    # the 4x3 hand KAT of SURVEY 8(a) a1
    Kk = np.array([[2., 0, 2], [0, 2, 1.5], [0, 0, 1]])
    pk = np.array([[0, -1, 0, 1], [1, 0, 0, 2], [0, 0, 1, 3]], dtype=np.float32)
    o, d = R.rays.make_o_d(4, 3, Kk, torch.from_numpy(pk))
    f1.update(kat_K=Kk, kat_pose=pk, kat_o=o, kat_d=d)
    save("F1_raygen", **f1)

    # ---- F2 ndc -------------------------------------------------------------------------------
    print("F2 ndc_rays")
    o, d = R.rays.make_o_d(Wf, Hf, Kf, torch.from_numpy(synthetic.fern_pose()[:3, :4]))
    sel = rs.choice(Hf * Wf, 256, replace=False)
    o_in, d_in = o.reshape(-1, 3)[sel].contiguous(), d.reshape(-1, 3)[sel].contiguous()
    o2, d2 = R.np.ndc_rays(Hf, Wf, Kf[0][0], 1.0, o_in, d_in)
    save("F2_ndc", H=Hf, W=Wf, focal=Kf[0][0], near=1.0, o_in=o_in, d_in=d_in, o_out=o2, d_out=d2)

    # ---- F3 stratified (via pre_process coarse branch) + F5 posenc / embed ---------------------
    print("F3 stratified, F5 posenc")
    enc_x, dim_x = R.posenc(10)
    enc_d, dim_d = R.posenc(4)
    assert (dim_x, dim_d) == (63, 27)
    o, d = R.rays.make_o_d(Wl, Hl, Kl, torch.from_numpy(synthetic.pose_spherical(0.0, -30.0, 4.0)[:3, :4]))
    pix = synthetic.pixel_batch(Hl, Wl, 4096, 0)
    rays_lego = torch.cat([o.reshape(-1, 3)[pix], d.reshape(-1, 3)[pix]], -1).contiguous()
    t64 = torch.from_numpy(counter_uniform(0, 0, 0, 64, 64))
    R.proxy.queue = [t64]
    emb, z, rd = R.np.pre_process(rays_lego[:64], (enc_x, enc_d), make_opts(), isFine=False)
    save("F3_stratified", near=2.0, far=6.0, t_rand=t64, z_vals=z)
    pts = torch.from_numpy(rs.uniform(-6, 6, size=(96, 3)).astype(np.float32))
    pts_ndc = torch.from_numpy(rs.uniform(-1.2, 1.2, size=(32, 3)).astype(np.float32))
    pts_all = torch.cat([pts, pts_ndc], 0)
    save("F5_posenc", pts=pts_all, enc10=enc_x(pts_all), enc4=enc_d(pts_all),
         rays8=rays_lego[:8], z8=z[:8], embedded8=emb[:8 * 64])

    # ---- F4 sample_pdf ------------------------------------------------------------------------
    print("F4 sample_pdf")
    n4 = 48
    zc = torch.sort(torch.from_numpy(rs.uniform(2, 6, size=(n4, 64)).astype(np.float32)), -1)[0]
    bins = 0.5 * (zc[:, 1:] + zc[:, :-1])
    w = torch.from_numpy((rs.uniform(0, 1, size=(n4, 62)) ** 6).astype(np.float32))
    w[0] = 0.0                                    # all-zero weights: uniform pdf
    w[1] = 0.0; w[1, 17] = 1.0                    # one-hot
    w[2, :40] = 0.0                               # long near-zero run (denom < 1e-5 branch)
    w[3] = 1e-7
    w[4, ::2] = 0.0
    u_inj = torch.from_numpy(counter_uniform(3, 1, 0, n4, 128))
    opts = make_opts()
    s_det = R.np.sample_pdf(bins, w, 128, det=True, opts=opts)
    R.proxy.queue = [u_inj]
    s_rand = R.np.sample_pdf(bins, w, 128, det=False, opts=opts)
    z_fine = torch.sort(torch.cat([zc, s_rand], -1), -1)[0]
    kat = R.np.sample_pdf(torch.linspace(2, 6, 5)[None], torch.tensor([[0.1, 0.0, 0.6, 0.3]]), 6, det=True, opts=opts)
    save("F4_sample_pdf", z_coarse=zc, bins=bins, weights=w, u=u_inj, samples_det=s_det, samples_rand=s_rand,
         z_fine=z_fine, kat_samples=kat)

    # ---- F6 MLP -------------------------------------------------------------------------------
    print("F6 MLP")
    x = emb[rs.choice(emb.shape[0], 384, replace=False)]
    xr = torch.from_numpy(rs.uniform(-1, 1, size=(128, 90)).astype(np.float32))
    x6 = torch.cat([x, xr], 0).contiguous()
    f6 = dict(x=x6, seed=7)
    for tag, (D, W) in (("d8w256", (8, 256)), ("d4w128", (4, 128))):
        m = ref_model(R, 7, D, W)
        f6[f"{tag}_coarse"] = m(x6)
        f6[f"{tag}_fine"] = m(x6, is_fine=True)
    save("F6_mlp", **f6)

    # ---- F7 post_process ----------------------------------------------------------------------
    print("F7 post_process")
    f7 = {}
    for S in (64, 192):
        n7 = 40
        raw = torch.from_numpy(rs.normal(0, 2, size=(n7, S, 4)).astype(np.float32))
        raw[..., 3] = raw[..., 3] * 8
        raw[0, :, 3] = -1.0                        # empty ray -> rgb 1, disp NaN->0, acc 0
        raw[1, :, 3] = 1e4                         # saturated at the first sample
        raw[2, :, 3] = 0.0; raw[2, S // 2, 3] = 50  # single surface
        raw[3, :, 3] = 1e-12                       # tiny density: acc ~ 0
        zz = torch.sort(torch.from_numpy(rs.uniform(2, 6, size=(n7, S)).astype(np.float32)), -1)[0]
        zz[4] = torch.linspace(0, 1, S)            # NDC-like range, depth/acc small -> disp clamp
        rd7 = torch.from_numpy(rs.normal(0, 1, size=(n7, 3)).astype(np.float32))
        rgb, disp, acc, wts, dep = R.np.post_process(raw, zz, rd7)
        f7.update({f"S{S}_raw": raw, f"S{S}_z": zz, f"S{S}_rays_d": rd7, f"S{S}_rgb": rgb, f"S{S}_disp": disp,
                   f"S{S}_acc": acc, f"S{S}_weights": wts, f"S{S}_depth": dep})
    rawk = torch.tensor([[[0, 0, 0, 1], [1, -1, 2, .5], [.5, .5, .5, -3], [2, 2, 2, 10]]], dtype=torch.float32)
    rgb, disp, acc, wts, dep = R.np.post_process(rawk, torch.tensor([[2, 3, 4.5, 6]]), torch.tensor([[0, 0, -2.]]))
    f7.update(kat_rgb=rgb, kat_disp=disp, kat_acc=acc, kat_weights=wts, kat_depth=dep)
    save("F7_post_process", **f7)

    # ---- F8 end-to-end render_rays with injected randoms ---------------------------------------
    print("F8 render_rays")
    o, d = R.rays.make_o_d(Wf, Hf, Kf, torch.from_numpy(synthetic.fern_pose()[:3, :4]))
    pixf = synthetic.pixel_batch(Hf, Wf, 64, 1)
    of, df = R.np.ndc_rays(Hf, Wf, Kf[0][0], 1.0, o.reshape(-1, 3)[pixf], d.reshape(-1, 3)[pixf])
    rays_fern = torch.cat([of, df], -1).contiguous()
    cases = {
        "legoA": dict(rays=rays_lego[:64], D=8, W=256, opts=make_opts()),
        "legoA_det": dict(rays=rays_lego[64:96], D=8, W=256, opts=make_opts(perturb=0.0)),
        "plumbP": dict(rays=rays_lego[:64], D=4, W=128, opts=make_opts(N_samples_f=0)),
        "fernN": dict(rays=rays_fern, D=8, W=256, opts=make_opts(near=0.0, far=1.0, data_type="llff")),
    }
    f8 = {}
    for tag, c in cases.items():
        m = ref_model(R, 0, c["D"], c["W"])
        n = c["rays"].shape[0]
        opts = c["opts"]
        t_rand = torch.from_numpy(counter_uniform(0, 0, 0, n, opts.N_samples_c))
        u = torch.from_numpy(counter_uniform(0, 1, 0, n, max(opts.N_samples_f, 1)))[:, :opts.N_samples_f]
        # staged capture: replay the same randoms through the reference's own stages
        R.proxy.queue = [t_rand]
        emb_c, z_c, rd = R.np.pre_process(c["rays"], (enc_x, enc_d), opts, isFine=False)
        raw_c = m(emb_c).reshape(n, opts.N_samples_c, 4)
        _, _, _, w_c, _ = R.np.post_process(raw_c, z_c, rd)
        f8.update({f"{tag}_rays": c["rays"], f"{tag}_t_rand": t_rand, f"{tag}_z_c": z_c, f"{tag}_raw_c": raw_c,
                   f"{tag}_weights_c": w_c, f"{tag}_D": c["D"], f"{tag}_W": c["W"],
                   f"{tag}_near": opts.near, f"{tag}_far": opts.far, f"{tag}_Nf": opts.N_samples_f,
                   f"{tag}_perturb": opts.perturb})
        if opts.N_samples_f > 0:
            R.proxy.queue = [] if opts.perturb == 0.0 else [u]
            emb_f, z_f, _ = R.np.pre_process(c["rays"], (enc_x, enc_d), opts, z_vals=z_c, weights=w_c, isFine=True)
            raw_f = m(emb_f, is_fine=True).reshape(n, -1, 4)
            f8.update({f"{tag}_u": u, f"{tag}_z_f": z_f, f"{tag}_raw_f": raw_f})
        # and the un-staged call
        R.proxy.queue = [t_rand] + ([] if (opts.N_samples_f == 0 or opts.perturb == 0.0) else [u])
        R.proxy.drawn = []
        out = R.np.render_rays(c["rays"], m, (enc_x, enc_d), opts)
        f8[f"{tag}_draws"] = np.array([len(s) and s[-1] for s in R.proxy.drawn])
        for k, v in out.items():
            f8[f"{tag}_{k}"] = v
    save("F8_render_rays", **f8)

    # ---- F9 batchify on a 16x16 image, chunk tail ----------------------------------------------
    print("F9 batchify")
    f9 = {}
    for tag, (K, H, W), pose, o9 in (
            ("blender", (Kl, Hl, Wl), synthetic.pose_spherical(40.0, -30.0, 4.0), make_opts(chunk_rays=100, N_samples_c=32, N_samples_f=64)),
            ("llff", (Kf, Hf, Wf), synthetic.fern_pose(), make_opts(chunk_rays=100, N_samples_c=32, N_samples_f=64, near=0.0, far=1.0, data_type="llff"))):
        # 16x16 crop-equivalent camera: same field of view, 16x16 pixels
        s = 16.0 / W
        K16 = K.copy(); K16[0, 0] *= s; K16[1, 1] *= s; K16[0, 2] = 8.0; K16[1, 2] = 8.0
        o, d = R.rays.make_o_d(16, 16, K16, torch.from_numpy(pose[:3, :4]))
        m = ref_model(R, 3, 4, 128)
        N = 256
        t_all = counter_uniform(5, 0, 0, N, 32)
        u_all = counter_uniform(5, 1, 0, N, 64)
        q = []
        for i in range(0, N, 100):
            q += [torch.from_numpy(t_all[i:i + 100]), torch.from_numpy(u_all[i:i + 100])]
        R.proxy.queue = q
        rc, dc, rf, df_ = R.np.batchify_rays_and_render_by_chunk(o, d, m, (enc_x, enc_d), 16, 16, K16, o9)
        f9.update({f"{tag}_K": K16, f"{tag}_pose": pose, f"{tag}_t_rand": t_all, f"{tag}_u": u_all,
                   f"{tag}_rgb_c": rc, f"{tag}_disp_c": dc, f"{tag}_rgb_f": rf, f"{tag}_disp_f": df_,
                   f"{tag}_near": o9.near, f"{tag}_far": o9.far})
    save("F9_batchify", **f9)
    f10_callers()
    f11_train_step(R)
    f12_llff_spiral()
    print("done")


if __name__ == "__main__":
    main()
