"""The callers either side of the path, kept on the device.

* ``test`` / ``render`` -- counterparts of the reference's eval and video harness (test.py:17-108, 111-174):
  checkpoint ingest (``model_state_dict`` of a reference ``.pth.tar``), one ``make_o_d`` +
  ``batchify_rays_and_render_by_chunk`` per pose, MSE/PSNR against ground truth, 8-bit frames.  Metrics and the
  8-bit conversion run as device kernels (``mi_nerf_image_metrics``, ``mi_nerf_nanmax``, ``mi_nerf_to8b``); the only
  device->host copies are the finished uint8 frames and two floats per frame, each copied ONCE after the last pose.  SSIM / LPIPS come from a third-party
  package the reference imports (IQA_pytorch) and are not part of this path: they are reported as ``None``.
* ``global_batch`` / ``GetterRayBatchIdx`` / ``ShuffledRows`` -- the global-batch ray precompute and epoch shuffle (main.py:92-106,
  utils.py:45-62) as one ray-generation launch over all training images; the shuffle is a permutation held beside the table
  and a step gathers its rows through it (no shuffled copy), instead of numpy on the host followed by a 2.3 GB upload.
* ``sample_rays_and_pixel`` -- per-image ray/pixel sampling (rays.py:36-64); rays are generated for the selected
  pixels only.
* ``spherical_poses`` (and ``get_render_pose`` / ``pose_spherical`` on top of it) -- the 360-degree camera path
  (dataset/render_pose.py:28-43): every pose of the path from one closed-form array expression, built on the host once and
  uploaded once.
* ``look_frames`` / ``rig_average`` / ``spiral_path`` / ``recenter_rig`` / ``llff_render_poses`` / ``llff_cameras`` -- the
  forward-facing (LLFF) camera path (dataset/load_llff.py:151-204, 277-346): the spiral the reference's loader hands to
  ``render()`` for ``data_type == 'llff'`` (main.py:43, test.py:125-145); all cameras of a path from batched array
  operations (120 poses of 15 numbers; one upload).

PNG output uses a 30-line zlib encoder (imageio, which the reference uses, is not a dependency of this path).
"""
from __future__ import annotations

import os
import struct
import zlib
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import nerf_process as NP
from . import ops
from ._lib import MiNerfError, as_f32_dev
from .rays import make_o_d


# ---------------------------------------------------------------------------------------------------
# small host utilities
# ---------------------------------------------------------------------------------------------------
def write_png(path: str, img: np.ndarray) -> None:
    """uint8 [H,W], [H,W,1] or [H,W,3] -> PNG (8-bit grey or RGB)."""
    a = np.ascontiguousarray(img)
    if a.dtype != np.uint8:
        raise MiNerfError(f"write_png wants uint8, got {a.dtype}")
    if a.ndim == 3 and a.shape[2] == 1:
        a = a[:, :, 0]
    if a.ndim == 2:
        color, raw = 0, a
    elif a.ndim == 3 and a.shape[2] == 3:
        color, raw = 2, a.reshape(a.shape[0], -1)
    else:
        raise MiNerfError(f"write_png: unsupported shape {a.shape}")
    h, w = a.shape[0], a.shape[1]
    rows = np.concatenate([np.zeros((h, 1), np.uint8), raw], axis=1).tobytes()        # filter byte 0 per scanline

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(rows, 6)) + chunk(b"IEND", b""))


def load_checkpoint(path: str, model: Optional[torch.nn.Module] = None, *, allow_pickle: bool = False) -> Dict:
    """torch.load of a reference checkpoint ({'idx', 'model_state_dict', 'optimizer_state_dict'}, train.py:105-114);
    loads ``model_state_dict`` into ``model`` when given (test.py:20-21).  The format holds tensors and plain containers only,
    so it is read with ``weights_only=True``; ``allow_pickle=True`` opts in to the unrestricted (code-executing) unpickler
    for checkpoints that need it."""
    ck = torch.load(path, map_location="cpu", weights_only=not allow_pickle)
    if "model_state_dict" not in ck:
        raise MiNerfError(f"{path}: no 'model_state_dict' (keys: {sorted(ck)[:8]})")
    if model is not None:
        model.load_state_dict(ck["model_state_dict"])
    return ck


def _ckpt_path(log_dir: str, exp_name: str, idx) -> str:
    return os.path.join(log_dir, exp_name, f"{exp_name}_{idx}.pth.tar")                 # test.py:20, train.py:112-114


# ---------------------------------------------------------------------------------------------------
# Camera paths, all poses of a path from one array expression (host side: a path is a few KB, built once and uploaded once).
# ---------------------------------------------------------------------------------------------------
def spherical_poses(theta_deg, phi_deg: float, radius: float) -> np.ndarray:
    """Camera-to-world matrices [N, 4, 4] (fp32) of cameras on a sphere of ``radius`` around the origin, looking at it, at azimuths
    ``theta_deg`` [N] and elevation ``phi_deg`` -- the closed form of the reference's matrix chain (dataset/render_pose.py:5-34:
    translate by radius along z, rotate about x by phi, about y by theta, swap axes).  Every entry of that chain is ONE fp32 product
    of fp32 sines / cosines (the other terms of each dot product are exact zeros), so writing the products out reproduces it bit
    for bit (fixture F10) without a matrix multiply per pose:

        c2w = [[-ct,  st*sp,  st*cp,  (st*cp)*r],
               [ st,  ct*sp,  ct*cp,  (ct*cp)*r],
               [  0,     cp,    -sp,      -sp*r],
               [  0,      0,      0,          1]]
    """
    th = np.atleast_1d(np.asarray(theta_deg, dtype=np.float64)) / 180.0 * np.pi
    ph = float(phi_deg) / 180.0 * np.pi
    ct, st = np.cos(th).astype(np.float32), np.sin(th).astype(np.float32)
    cp, sp, r = np.float32(np.cos(ph)), np.float32(np.sin(ph)), np.float32(radius)
    zero, one = np.zeros_like(ct), np.ones_like(ct)
    fwd = np.stack([st * cp, ct * cp, -sp * one], -1)                 # third column: where the camera sits, per unit radius
    rows = np.stack([np.stack([-ct, st * sp, fwd[:, 0], fwd[:, 0] * r], -1),
                     np.stack([st, ct * sp, fwd[:, 1], fwd[:, 1] * r], -1),
                     np.stack([zero, cp * one, fwd[:, 2], fwd[:, 2] * r], -1),
                     np.stack([zero, zero, zero, one], -1)], 1)
    return rows.astype(np.float32)


def pose_spherical(theta: float, phi: float, radius: float) -> torch.Tensor:
    """One pose of ``spherical_poses`` (dataset/render_pose.py:28-34) -> [4, 4]."""
    return torch.from_numpy(spherical_poses([theta], phi, radius)[0])


def get_render_pose(n_angle: int = 1, single_angle: float = -1, phi: float = -30.0, nf: float = 4.0, device=None) -> torch.Tensor:
    """The video path of blender / custom scenes (dataset/render_pose.py:37-43) -> [n, 4, 4]: ``n_angle`` azimuths evenly over the
    circle starting at -180 degrees, or the one view ``single_angle`` when that is given (or when only one view is asked for).
    Uploaded once when ``device`` is given."""
    circle = n_angle != 1 and single_angle == -1
    azimuths = np.linspace(-180, 180, n_angle + 1)[:-1] if circle else [single_angle]
    poses = torch.from_numpy(spherical_poses(azimuths, phi, nf))
    return poses.to(device) if device is not None else poses


# LLFF (forward-facing) rigs: poses are [N, 3, 5] camera-to-world matrices with the (H, W, focal) column of poses_bounds.npy appended
# (dataset/load_llff.py:151-204; the spiral set-up of :277-346).  Arithmetic stays in the dtype the loader has at that point (the rig is
# fp32 once loaded, the spiral is float64): the fixtures pin these paths bit for bit.  The last axis is the vector axis.
def _unit(v: np.ndarray) -> np.ndarray:
    """Vectors scaled to unit length along the last axis (one vector: numpy's own 2-norm, so an fp32 vector rounds as it does in the loader)."""
    if v.ndim == 1:
        return v / np.linalg.norm(v)
    return v / np.sqrt(np.square(v).sum(-1, keepdims=True))


def look_frames(back: np.ndarray, up: np.ndarray, eye: np.ndarray) -> np.ndarray:
    """Camera frames [..., 3, 4] = (right, up', back, eye) as columns for cameras at ``eye`` whose z axis is ``back`` and whose y axis is as
    close to ``up`` as orthogonality allows (load_llff.py:155-161, for any number of cameras at once; ``up`` broadcasts)."""
    z = _unit(np.asarray(back))
    x = _unit(np.cross(np.broadcast_to(up, z.shape), z))
    y = _unit(np.cross(z, x))
    return np.stack([x, y, z, np.broadcast_to(eye, z.shape)], -1)


def rig_average(poses: np.ndarray) -> np.ndarray:
    """The average camera of a rig [3, 5] (load_llff.py:169-176): mean position, summed viewing and up directions, hwf of camera 0."""
    # the summed viewing direction is made a unit vector BEFORE the frame is built (which scales it again): in fp32 the second pass can
    # move an ulp, and the loader does both
    frame = look_frames(_unit(poses[:, :3, 2].sum(0)), poses[:, :3, 1].sum(0), poses[:, :3, 3].mean(0))
    return np.concatenate([frame, poses[0, :3, -1:]], 1)


def spiral_path(c2w: np.ndarray, up: np.ndarray, radii, focus: float, zrate: float, rots: float, n: int) -> np.ndarray:
    """``n`` cameras [n, 3, 5] on a spiral around the camera ``c2w`` [3, 5], all looking at the point ``focus`` in front of it
    (load_llff.py:179-189).  In c2w's own frame camera k sits at radii * (cos t, -sin t, -sin(zrate t)), t = 2 pi rots k / n."""
    t = np.linspace(0.0, 2.0 * np.pi * rots, int(n) + 1)[:-1]
    frame = c2w[:3, :4]
    local = np.stack([np.cos(t), -np.sin(t), -np.sin(t * zrate), np.ones_like(t)], -1) * np.append(np.asarray(radii, dtype=np.float64), 1.0)
    eyes = (local[:, None, :] * frame[None]).sum(-1)                  # frame @ local_k for every k
    target = (frame * np.array([0.0, 0.0, -focus, 1.0])).sum(-1)
    cams = look_frames(eyes - target, up, eyes)
    return np.concatenate([cams, np.broadcast_to(c2w[:, 4:5], (len(t), 3, 1))], -1)


def recenter_rig(poses: np.ndarray) -> np.ndarray:
    """Every camera of the rig expressed in the frame of the rig's average camera (load_llff.py:192-204)."""
    last_row = np.array([0.0, 0.0, 0.0, 1.0])
    world_from_avg = np.vstack([rig_average(poses)[:3, :4], last_row])
    homog = np.concatenate([poses[:, :3, :4], np.broadcast_to(last_row, (poses.shape[0], 1, 4))], 1)
    out = poses + 0
    out[:, :3, :4] = (np.linalg.inv(world_from_avg) @ homog)[:, :3, :4]
    return out


def llff_render_poses(poses, bds, path_zflat: bool = False, n_views: int = 120, n_rots: int = 2) -> np.ndarray:
    """The spiral the loader derives from the recentred rig and its depth bounds (load_llff.py:294-328) -> fp32 [n, 3, 5]: centred on the
    rig's average camera, focused at a depth 3/4 of the way (in disparity) from 0.9 x the nearest to 5 x the farthest bound, with the
    90th percentile of the cameras' |offsets| as radii.  ``path_zflat``: a flat one-turn path slightly behind the rig; the reference
    halves ``N_views`` with a true division and then fails inside np.linspace on current numpy (load_llff.py:321,184); here the count
    stays an integer."""
    centre = rig_average(poses)
    up = _unit(poses[:, :3, 1].sum(0))
    nearest, farthest = bds.min() * .9, bds.max() * 5.
    focus = 1. / (.25 / nearest + .75 / farthest)
    radii = np.percentile(np.abs(poses[:, :3, 3]), 90, 0)
    if path_zflat:
        centre[:3, 3] = centre[:3, 3] - nearest * .1 * centre[:3, 2]
        radii[2] = 0.
        n_rots, n_views = 1, n_views // 2
    return spiral_path(centre, up, radii, focus, zrate=.5, rots=n_rots, n=n_views).astype(np.float32)


def llff_cameras(raw_poses, raw_bds, bd_factor=.75, path_zflat: bool = False) -> Dict:
    """Everything load_llff() derives from ``poses_bounds.npy`` (raw_poses [3,5,N], raw_bds [2,N]; load_llff.py:277-346) except
    the images: axis reorder, rescale by the near bound, recentring, the render spiral, intrinsics.  Returns a dict with
    ``poses`` [N,3,5], ``bds`` [N,2], ``render_poses`` [120,3,5], ``gt_extrinsic`` [N,3,4], ``gt_intrinsic`` [3,3], ``hw``."""
    # (x, y, z) columns of poses_bounds are (down, right, back): reorder to (right, up, back); cameras first
    poses = np.moveaxis(np.concatenate([raw_poses[:, 1:2, :], -raw_poses[:, 0:1, :], raw_poses[:, 2:, :]], 1), -1, 0).astype(np.float32)
    bds = np.moveaxis(raw_bds, -1, 0).astype(np.float32)
    scale = 1. if bd_factor is None else 1. / (bds.min() * bd_factor)
    poses[:, :3, 3] *= scale
    bds *= scale
    poses = recenter_rig(poses)
    render_poses = llff_render_poses(poses, bds, path_zflat)
    poses = poses.astype(np.float32)
    H, W, focal = poses[0, :3, -1]
    H, W = int(H), int(W)
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    return {"poses": poses, "bds": bds, "render_poses": render_poses, "gt_extrinsic": poses[:, :3, :4], "gt_intrinsic": K, "hw": [H, W]}


# ---------------------------------------------------------------------------------------------------
# eval / video harness
# ---------------------------------------------------------------------------------------------------
def _precision(opts) -> Dict[str, bool]:
    """``opts.precision`` (not a flag of the reference's config.py; absent = "fp32", the reference's arithmetic) selects the network's
    precision mode for the eval / video harness: "fp32" | "f16s" (split precision: fp32-grade results, ~3x faster) | "bf16" |
    "f16s+bf16" (coarse network in split precision -- the fine sample positions are the fp32 path's -- fine network in bf16)."""
    mode = str(getattr(opts, "precision", "fp32")).lower()
    if mode not in ("fp32", "f16s", "bf16", "f16s+bf16"):
        raise ValueError(f"opts.precision must be 'fp32', 'f16s', 'bf16' or 'f16s+bf16', got {mode!r}")
    return {"bf16": mode in ("bf16", "f16s+bf16"), "f16s": mode == "f16s", "coarse_f16s": mode == "f16s+bf16"}


def _frozen(model, opts):
    """The model packed ONCE for a whole test() / render() call (the weights do not change inside it; packed_for() on an nn.Module re-packs
    per call otherwise).  Split precision: the blobs are built here and the device packer's out-of-range count is read once -- a checkpoint
    with a weight beyond the f16 range raises instead of rendering from a clipped network."""
    from .weights import packed_for
    packed = packed_for(model)
    if _precision(opts)["f16s"] or _precision(opts)["coarse_f16s"]:
        packed.f16s()
        packed.check_f16s_range()
    return packed


def _render_pose(model, posenc, K, pose, hw, opts):
    img_h, img_w = hw
    rays_o, rays_d = make_o_d(img_w, img_h, K, pose[:3, :4])
    rgb_c, disp_c, rgb_f, disp_f = NP.batchify_rays_and_render_by_chunk(rays_o, rays_d, model, posenc, img_h, img_w, K, opts, **_precision(opts))
    return (rgb_c, disp_c) if int(opts.N_samples_f) == 0 else (rgb_f, disp_f)             # test.py:42-47


def test(idx, i_test, posenc, model, test_imgs, gt_intrinsic, gt_extrinsic, hw, opts, *, log_dir: Optional[str] = None,
         save_dir: Optional[str] = None, keep_frames: bool = False) -> Dict:
    """Counterpart of test.py:17-108.  ``log_dir`` given: load ``{log_dir}/{exp_name}/{exp_name}_{idx}.pth.tar`` first
    (test.py:20-21).  ``save_dir`` given: write ``NNN.png``, ``NNN_disp.png`` and ``_result.txt`` like the reference.
    Returns per-frame loss / PSNR (Python floats), best / mean PSNR, and the uint8 frames when ``keep_frames``."""
    if isinstance(model, torch.nn.Module):
        model.eval()
        if log_dir is not None:
            load_checkpoint(_ckpt_path(log_dir, opts.exp_name, idx), model)
    if save_dir is not None:
        os.makedirs(save_dir, exist_ok=True)
    img_h, img_w = hw
    dev = next(model.parameters()).device if isinstance(model, torch.nn.Module) else model.device
    n_pose = len(gt_extrinsic)
    want_frames = save_dir is not None or keep_frames
    # everything a frame produces stays on the device until the loop is over: [mse, psnr] per frame in one [N, 2] tensor, the uint8
    # frames in two stacks -- ONE device -> host copy each after the last pose, no host synchronisation per frame
    metrics = torch.empty(n_pose, 2, dtype=torch.float32, device=dev)
    rgbs8 = torch.empty(n_pose, img_h, img_w, 3, dtype=torch.uint8, device=dev) if want_frames else None
    disps8 = torch.empty(n_pose, img_h, img_w, 1, dtype=torch.uint8, device=dev) if want_frames else None
    with torch.no_grad():
        frozen = _frozen(model, opts)
        for i, pose in enumerate(gt_extrinsic):
            pose = as_f32_dev(pose, dev)
            pred_rgb, pred_disp = _render_pose(frozen, posenc, gt_intrinsic, pose, hw, opts)
            target = as_f32_dev(test_imgs[i], pred_rgb.device).reshape(-1, 3)             # test.py:63
            metrics[i] = ops.image_metrics(pred_rgb, target)                              # img2mse, mse2psnr: test.py:65-67
            if want_frames:
                rgbs8[i] = ops.to8b(pred_rgb).reshape(img_h, img_w, 3)                        # test.py:55
                disps8[i] = ops.to8b(pred_disp, ops.nanmax(pred_disp)).reshape(img_h, img_w, 1)   # test.py:56
    m_host = metrics.cpu().numpy()
    losses: List[float] = [float(v) for v in m_host[:, 0]]
    psnrs: List[float] = [float(v) for v in m_host[:, 1]]
    frames = []
    if want_frames:
        rgbs_np, disps_np = rgbs8.cpu().numpy(), disps8.cpu().numpy()
        for i in range(n_pose):
            if save_dir is not None:
                write_png(os.path.join(save_dir, f"{i:03d}.png"), rgbs_np[i])
                write_png(os.path.join(save_dir, f"{i:03d}_disp.png"), disps_np[i])
            if keep_frames:
                frames.append((rgbs_np[i], disps_np[i]))
    best = int(np.argmax(psnrs)) if psnrs else -1
    res = {"loss": losses, "psnr": psnrs, "ssim": None, "lpips": None, "best_idx": best,
           "best_psnr": psnrs[best] if psnrs else None, "mean_psnr": float(np.mean(psnrs)) if psnrs else None}
    if keep_frames:
        res["frames"] = frames
    if save_dir is not None:                                                              # test.py:92-108
        with open(os.path.join(save_dir, "_result.txt"), "w") as f:
            for i in range(len(losses)):
                f.write(f"idx:{i}\tloss:{losses[i]}\tpsnr:{psnrs[i]}\tssim:n/a\tlpips:n/a\n")
            f.write(f"\nBest Value ) PSNR : {res['best_psnr']}\tSSIM : n/a\tLPIPS : n/a\n")
            f.write(f"Mean Value ) PSNR : {res['mean_psnr']}\tSSIM : n/a\tLPIPS : n/a")
    return res


def render(idx, posenc, model, gt_intrinsic, render_pose, hw, opts, *, log_dir: Optional[str] = None, save_dir: Optional[str] = None):
    """Counterpart of test.py:111-174: renders every pose of ``render_pose`` -- for blender/custom data the spherical path
    from opts (test.py:119-124); for llff data the spiral the loader produced (``llff_render_poses``; [n,3,5] or [n,4,4],
    the first 3x4 block is the camera) -- and returns ``(rgbs uint8 [N,H,W,3], disps uint8 [N,H,W])`` -- the arrays the reference
    hands to imageio.mimwrite (test.py:166-172).  ``save_dir``: also write ``{i}_rgb.png`` / ``{i}_disp.png``."""
    dev = next(model.parameters()).device if isinstance(model, torch.nn.Module) else model.device
    if getattr(opts, "data_type", None) in ("blender", "custom"):
        render_pose = get_render_pose(n_angle=opts.n_angle, single_angle=opts.single_angle, phi=opts.phi, nf=opts.nf)
    elif render_pose is None:
        raise MiNerfError("render(): llff data needs the loader's render poses (harness.llff_render_poses); got None")
    poses = torch.as_tensor(np.asarray(render_pose) if not isinstance(render_pose, torch.Tensor) else render_pose, dtype=torch.float32).to(dev)
    if isinstance(model, torch.nn.Module):
        model.eval()
        if log_dir is not None:
            load_checkpoint(_ckpt_path(log_dir, opts.exp_name, idx), model)
    if save_dir is not None:
        os.makedirs(save_dir, exist_ok=True)
    img_h, img_w = hw
    n = poses.shape[0]
    rgbs = torch.empty(n, img_h, img_w, 3, dtype=torch.uint8, device=dev)
    disps = torch.empty(n, img_h, img_w, dtype=torch.uint8, device=dev)
    with torch.no_grad():
        frozen = _frozen(model, opts)
        for i in range(n):
            rgb, disp = _render_pose(frozen, posenc, gt_intrinsic, poses[i], hw, opts)
            rgbs[i] = ops.to8b(rgb).reshape(img_h, img_w, 3)                              # to8b(rgbs), test.py:167
            disps[i] = ops.to8b(disp, ops.nanmax(disp)).reshape(img_h, img_w)             # disp / nanmax, test.py:156,168
    rgbs_np, disps_np = rgbs.cpu().numpy(), disps.cpu().numpy()                           # one copy for the whole clip
    if save_dir is not None:
        for i in range(n):
            write_png(os.path.join(save_dir, f"{i}_rgb.png"), rgbs_np[i])
            write_png(os.path.join(save_dir, f"{i}_disp.png"), disps_np[i])
    return rgbs_np, disps_np


# ---------------------------------------------------------------------------------------------------
# training data staging
# ---------------------------------------------------------------------------------------------------
class ShuffledRows:
    """``table[perm]`` without the copy: what ``GetterRayBatchIdx`` hands out in place of the reference's shuffled ``rays_rgb`` array.
    The reference shuffles the whole table in memory (main.py:102, utils.py:49-50) and then reads B consecutive rows per step
    (train.py:29).  A random 36-byte row costs a 128-byte HBM request, so materialising the shuffle moves ~2.5x its algorithmic
    bytes and needs a second 2.3 GB table (100 views of 800x800); here only the permutation exists, and ``rows[a:b]`` -- the one
    access the training loop makes -- gathers those b - a rows from the unshuffled table in one small launch (ops.gather_rows).
    ``materialize()`` is the reference's array, for whoever wants it whole."""

    def __init__(self, table: torch.Tensor, perm: torch.Tensor):
        self.table, self.perm = table, perm

    @property
    def shape(self):
        return self.table.shape

    @property
    def device(self):
        return self.table.device

    def __len__(self) -> int:
        return int(self.table.shape[0])

    def __getitem__(self, key) -> torch.Tensor:
        if not isinstance(key, slice):
            raise MiNerfError("a shuffled global batch is read in slices of consecutive rows (train.py:29); materialize() gives the whole array")
        return ops.gather_rows(self.table, self.perm[key].contiguous())

    def materialize(self) -> torch.Tensor:
        return ops.permute_rows(self.table, self.perm)


class GetterRayBatchIdx:
    """Epoch cursor over the shuffled global batch (utils.py:45-62), on the device.  ``__call__(batch)`` returns
    ``(i_batch, rays_rgb, epoch)`` like the reference; the caller slices ``rays_rgb[i_batch - B : i_batch]`` (train.py:27-29).
    ``rays_rgb`` is the table itself while nothing has been shuffled, else a ``ShuffledRows`` view of it: a reshuffle
    (construction with ``shuffle=True``, and every epoch end, utils.py:47-52) draws a new permutation -- 8 bytes per ray -- and
    moves no ray."""

    def __init__(self, rays_rgb: torch.Tensor, generator: Optional[torch.Generator] = None, shuffle: bool = False):
        self.table = rays_rgb
        self.perm: Optional[torch.Tensor] = None
        self.epoch = 0
        self.i_batch = 0
        self._gen = generator
        if shuffle:
            self._draw()

    def _draw(self) -> None:
        self.perm = torch.randperm(self.table.shape[0], device=self.table.device, generator=self._gen)     # utils.py:49 / main.py:102

    @property
    def rays_rgb(self):
        return self.table if self.perm is None else ShuffledRows(self.table, self.perm)

    def shuffle_ray_idx(self, batch_size: int) -> None:
        # the reference permutes the already shuffled array again (utils.py:49-50): a uniform permutation of a permutation is a uniform
        # permutation, so a fresh draw has the same distribution
        self._draw()
        self.i_batch = batch_size
        self.epoch += 1

    def __call__(self, batch_size: int):
        self.i_batch += batch_size
        if self.i_batch >= self.table.shape[0]:
            self.shuffle_ray_idx(batch_size)
        return self.i_batch, self.rays_rgb, self.epoch


def global_batch(images, gt_intrinsic, gt_extrinsic, i_train: Sequence[int], hw, device, generator: Optional[torch.Generator] = None,
                 shuffle: bool = True) -> GetterRayBatchIdx:
    """main.py:92-106: rays for every training image, concatenated with the pixels, flattened to [N*H*W, 3, 3], shuffled (as a
    permutation held beside the table: ``ShuffledRows``).
    ``images`` [n_img,H,W,3] and ``gt_extrinsic`` [n_img,4,4] may live on the host: only the selected views are uploaded."""
    img_h, img_w = hw
    idx = torch.as_tensor(np.asarray(list(i_train)), dtype=torch.long)
    imgs = torch.as_tensor(np.asarray(images) if not isinstance(images, torch.Tensor) else images)
    poses = torch.as_tensor(np.asarray(gt_extrinsic) if not isinstance(gt_extrinsic, torch.Tensor) else gt_extrinsic)
    imgs = imgs[idx.to(imgs.device)].to(device=device, dtype=torch.float32).contiguous()
    poses = poses[idx.to(poses.device)][:, :3, :4].to(device=device, dtype=torch.float32).contiguous()
    rr = ops.rays_rgb(img_w, img_h, gt_intrinsic, poses, imgs)
    return GetterRayBatchIdx(rr, generator, shuffle=shuffle)                                            # main.py:102: a permutation, not a copy


def sample_rays_and_pixel(i, img_w, img_h, K, pose, target_img, opts, generator: Optional[torch.Generator] = None):
    """rays.py:36-64 fused with the ``make_o_d`` that precedes it in train.py:43-45: N_rays pixels without replacement
    (inside the centre crop while ``i < opts.precrop_iters``), rays generated for those pixels only.
    Returns ``(rays_o [N,3], rays_d [N,3], target [N,3])``."""
    target_img = as_f32_dev(target_img)
    dev = target_img.device
    if i < int(getattr(opts, "precrop_iters", 0)):
        dH = int(img_h // 2 * opts.precrop_frac)
        dW = int(img_w // 2 * opts.precrop_frac)
        ys = torch.arange(img_h // 2 - dH, img_h // 2 + dH, device=dev)
        xs = torch.arange(img_w // 2 - dW, img_w // 2 + dW, device=dev)
    else:
        ys = torch.arange(img_h, device=dev)
        xs = torch.arange(img_w, device=dev)
    n_coords = ys.numel() * xs.numel()
    n = int(opts.N_rays)
    if n > n_coords:
        raise MiNerfError(f"cannot draw {n} pixels without replacement from {n_coords}")       # np.random.choice raises too
    sel = torch.randperm(n_coords, device=dev, generator=generator)[:n]                         # rays.py:53-54
    py, px = ys[sel // xs.numel()], xs[sel % xs.numel()]
    pix = (py * img_w + px).contiguous()
    rays_o, rays_d = ops.make_o_d_pixels(int(img_w), int(img_h), K, pose, pix)
    target = target_img.reshape(-1, 3)[pix]
    return rays_o, rays_d, target


def train(idx, i_train, images, gt_cam_param, hw, model, criterion, posenc, optimizer, global_batch_idx, vis, opts, *,
          log_dir: Optional[str] = None, generator: Optional[torch.Generator] = None) -> Dict:
    """Counterpart of train.py:12-119 with the reference's signature: one optimisation step through the training path.
    Ray/pixel selection: the global-batch cursor (train.py:25-32) or per-image sampling (train.py:35-45, here without the
    full-frame ``make_o_d``); render with gradients (train.py:53); ``criterion`` on the coarse and fine colours
    (train.py:60-66); ``loss.backward(); optimizer.step()`` (train.py:69-70); checkpoint in the reference's format every
    ``opts.idx_save`` steps when ``log_dir`` is given (train.py:105-114).  ``vis`` (visdom) is accepted and unused.
    Returns the losses and PSNRs as 0-dim device tensors (no host synchronisation).  With ``opts.precision == "f16s"`` the dict also
    carries ``f16s`` = train_path.f16s_status(model) every ``opts.idx_print`` steps (one device -> host read): the share of the f16 range
    the scaled backward used and the packer's out-of-range count; the training path itself raises when either says a step was clipped."""
    model.train()
    img_h, img_w = hw
    gt_intrinsic, gt_extrinsic = gt_cam_param
    dev = next(model.parameters()).device
    if global_batch_idx is not None and getattr(opts, "global_batch", True):
        i_batch, rays_rgb, _ = global_batch_idx(int(opts.N_rays))
        batch = rays_rgb[i_batch - int(opts.N_rays):i_batch]                                  # [B, 3, 3]   train.py:29
        rays_o, rays_d, target_img = batch[:, 0], batch[:, 1], batch[:, 2]
    else:
        i_img = int(np.random.choice(i_train))                                                # train.py:37
        target_full = as_f32_dev(torch.as_tensor(images[i_img]), dev)
        pose = torch.as_tensor(np.asarray(gt_extrinsic[i_img]) if not isinstance(gt_extrinsic, torch.Tensor) else gt_extrinsic[i_img])
        rays_o, rays_d, target_img = sample_rays_and_pixel(idx, img_w, img_h, gt_intrinsic, pose[:3, :4], target_full, opts, generator)
    rgb_c, _, rgb_f, _ = NP.batchify_rays_and_render_by_chunk(rays_o.contiguous(), rays_d.contiguous(), model, posenc, img_h, img_w,
                                                              gt_intrinsic, opts, **_precision(opts))      # train.py:53
    optimizer.zero_grad()
    target_img = target_img.contiguous()
    loss = criterion(rgb_c, target_img)                                                       # train.py:60
    out = {"loss_c": loss.detach(), "psnr_c": -10.0 * torch.log10(loss.detach())}
    if int(opts.N_samples_f) > 0:
        loss_f = criterion(rgb_f, target_img)
        out.update(loss_f=loss_f.detach(), psnr_f=-10.0 * torch.log10(loss_f.detach()))
        loss = loss + loss_f                                                                  # train.py:66
    out["loss"] = loss.detach()
    loss.backward()                                                                           # train.py:69
    optimizer.step()                                                                          # train.py:70
    if _precision(opts)["f16s"] and idx % int(getattr(opts, "idx_print", 0) or 100) == 0:
        from . import train_path
        out["f16s"] = train_path.f16s_status(model, reset=False)
    if log_dir is not None and idx % int(getattr(opts, "idx_save", 0) or (1 << 62)) == 0 and idx > 0:
        save_path = os.path.join(log_dir, opts.exp_name)
        os.makedirs(save_path, exist_ok=True)
        torch.save({"idx": idx, "model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict()},
                   _ckpt_path(log_dir, opts.exp_name, idx))                                   # train.py:105-114
    return out
