"""CPU-side checks of the harness module: camera path against the reference's own
output (fixture F10), the PNG encoder, checkpoint ingest in the reference's format.  No GPU compute."""
import os
import struct
import zlib

import numpy as np
import pytest
import torch

from nerf_pytorch_paeng_amd import harness, synthetic
from nerf_pytorch_paeng_amd._lib import MiNerfError
from nerf_pytorch_paeng_amd.model import NeRF


def test_render_pose_matches_reference(golden):
    """All poses of a spherical path come from ONE closed-form array expression (harness.spherical_poses): bit for bit the reference's
    per-pose matrix chain (fixture F10), for a whole circle, a single view, and any subset of azimuths at once."""
    g = golden("F10_callers")
    np.testing.assert_array_equal(harness.spherical_poses(np.linspace(-180, 180, 9)[:-1], -30.0, 4.0), g["poses8"])
    np.testing.assert_array_equal(harness.spherical_poses([33.0, 120.0], -41.0, 4.0)[0], g["pose_sph"])
    assert harness.spherical_poses(np.linspace(-180, 180, 41)[:-1], -30.0, 4.0).shape == (40, 4, 4)
    np.testing.assert_array_equal(harness.get_render_pose(n_angle=8, single_angle=-1, phi=-30.0, nf=4.0).numpy(), g["poses8"])
    np.testing.assert_array_equal(harness.get_render_pose(n_angle=1, single_angle=120, phi=-20.0, nf=3.5).numpy(), g["pose_single"])
    np.testing.assert_array_equal(harness.pose_spherical(33.0, -41.0, 4.0).numpy(), g["pose_sph"])


def _decode_png(path):
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, {}
    while pos < len(b):
        n, tag = struct.unpack(">I4s", b[pos:pos + 8])
        data = b[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", b[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + data) & 0xFFFFFFFF
        chunks.setdefault(tag, b"")
        chunks[tag] += data
        pos += 12 + n
    w, h, depth, color = struct.unpack(">IIBB", chunks[b"IHDR"][:10])
    ch = {0: 1, 2: 3}[color]
    raw = np.frombuffer(zlib.decompress(chunks[b"IDAT"]), np.uint8).reshape(h, 1 + w * ch)
    assert depth == 8 and (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, ch)


@pytest.mark.parametrize("shape", [(5, 7, 3), (4, 6), (3, 3, 1)])
def test_png_roundtrip(tmp_path, shape):
    img = np.random.RandomState(0).randint(0, 256, shape).astype(np.uint8)
    p = str(tmp_path / "a.png")
    harness.write_png(p, img)
    np.testing.assert_array_equal(_decode_png(p).reshape(img.shape), img)
    with pytest.raises(MiNerfError):
        harness.write_png(p, img.astype(np.float32))


def test_checkpoint_ingest_reference_format(tmp_path):
    """train.py:105-114 saves {'idx', 'model_state_dict', 'optimizer_state_dict'}; test.py:20-21 loads it."""
    src = NeRF(4, 128, 63, 27)
    src.load_state_dict({k: torch.as_tensor(v) for k, v in synthetic.make_state_dict(9, 4, 128).items()})
    opt = torch.optim.Adam(src.parameters(), lr=5e-4)
    exp = "lego_exp"
    os.makedirs(tmp_path / exp)
    path = harness._ckpt_path(str(tmp_path), exp, 1000)
    torch.save({"idx": 1000, "model_state_dict": src.state_dict(), "optimizer_state_dict": opt.state_dict()}, path)
    assert path.endswith(os.path.join(exp, "lego_exp_1000.pth.tar"))
    dst = NeRF(4, 128, 63, 27)
    ck = harness.load_checkpoint(path, dst)
    assert ck["idx"] == 1000
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), k
    torch.save({"idx": 1}, path)
    with pytest.raises(MiNerfError):
        harness.load_checkpoint(path)


def test_bench_launcher_propagates_worker_failure():
    """`python bench.py --gpus 2` without torchrun starts its own workers BEFORE touching the GPU; here (no GPU) both workers
    exit with an error and the launcher must report it instead of hanging or printing a result."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-only check of the launcher's error path")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs an MI355X" in r.stderr and "worker rank" in r.stderr
    assert not any(l.startswith("{") for l in r.stdout.splitlines())


def test_F12_llff_spiral_path(golden):
    """The LLFF camera path (dataset/load_llff.py:151-204, 277-346) against outputs of the reference's own load_llff() run on
    synthetic poses_bounds (fixture F12): render spiral, recentred rig, intrinsics -- bit for bit."""
    import numpy as np
    from nerf_pytorch_paeng_amd import harness as Hn
    g = golden("F12_llff_spiral")
    cam = Hn.llff_cameras(g["raw_poses"], g["raw_bds"])
    np.testing.assert_array_equal(cam["render_poses"], g["spiral_render_poses"])
    assert cam["render_poses"].shape == (120, 3, 5) and cam["render_poses"].dtype == np.float32
    np.testing.assert_array_equal(cam["gt_extrinsic"], g["spiral_extrinsic"])
    np.testing.assert_array_equal(cam["gt_intrinsic"], g["spiral_K"])
    np.testing.assert_array_equal(cam["poses"], g["rec_poses"])
    np.testing.assert_array_equal(cam["bds"], g["rec_bds"])
    assert cam["hw"] == [int(v) for v in g["hw"]]
    # building blocks.  The loader's arrays keep the memory order of poses_bounds ([3,5,N] moved to [N,3,5] as a view), and
    # numpy's reductions over axis 0 round differently for different strides: rebuild that layout for the bit-exact check
    rec = np.moveaxis(np.ascontiguousarray(np.moveaxis(g["rec_poses"], 0, -1)), -1, 0)
    np.testing.assert_array_equal(Hn.rig_average(rec), g["poses_avg"])
    np.testing.assert_allclose(Hn.rig_average(g["rec_poses"]), g["poses_avg"], atol=1e-7)
    np.testing.assert_array_equal(Hn._unit(g["normalize_in"]), g["normalize_out"])
    np.testing.assert_array_equal(Hn.look_frames(rec[0, :3, 2], rec[1, :3, 1], rec[2, :3, 3]), g["viewmatrix_out"])
    # all nine cameras of a spiral from ONE array expression: float64, the reference's per-camera loop agrees to the last-but-one bit
    # (sums of four products in another association) and exactly once rounded to the fp32 the loader hands out
    sp = Hn.spiral_path(g["poses_avg"], Hn._unit(rec[:, :3, 1].sum(0)), np.array([0.3, 0.2, 0.1]), 3.5, zrate=.5, rots=2, n=9)
    assert sp.shape == (9, 3, 5) and sp.dtype == np.float64
    np.testing.assert_allclose(sp, g["spiral_direct"], rtol=0, atol=5e-16)
    np.testing.assert_array_equal(sp.astype(np.float32), g["spiral_direct"].astype(np.float32))
    # a batch of frames equals the frames one by one
    many = Hn.look_frames(rec[:, :3, 2].astype(np.float64), rec[1, :3, 1].astype(np.float64), rec[:, :3, 3].astype(np.float64))
    for i in range(rec.shape[0]):
        np.testing.assert_allclose(many[i], Hn.look_frames(rec[i, :3, 2].astype(np.float64), rec[1, :3, 1].astype(np.float64), rec[i, :3, 3].astype(np.float64)),
                                   rtol=0, atol=5e-16)
    np.testing.assert_array_equal(Hn.recenter_rig(rec), Hn.recenter_rig(rec))
    # properties: every spiral camera is orthonormal and looks at the focus point; the flat path halves the view count
    Rm = cam["render_poses"][:, :3, :3].astype(np.float64)
    np.testing.assert_allclose(np.einsum("nij,nik->njk", Rm, Rm), np.broadcast_to(np.eye(3), (120, 3, 3)), atol=1e-6)
    assert Hn.llff_render_poses(cam["poses"], cam["bds"], path_zflat=True).shape == (60, 3, 5)
