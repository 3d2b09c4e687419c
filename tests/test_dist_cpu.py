"""Multi-process sharding logic on CPU (gloo, world_size 2): row sharding, ragged tile gather, and a whole
sharded frame assembled from per-rank tiles equals the single-process frame.  The per-rank renderer here is
the CPU oracle (test infrastructure); on the GPU box the same dist.render_frame drives the HIP path."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_rows_partition():
    from nerf_pytorch_paeng_amd.dist import shard_rows
    for H in (800, 378, 7, 3):
        for world in (1, 2, 3, 4, 8):
            blocks = [shard_rows(H, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and sum(n for _, n in blocks) == H
            for (r0, n0), (r1, _) in zip(blocks, blocks[1:]):
                assert r0 + n0 == r1
            assert max(n for _, n in blocks) - min(n for _, n in blocks) <= 1
    assert shard_rows(800, 8, 3) == (300, 100)
    assert [shard_rows(378, 8, r)[1] for r in range(8)] == [48, 48, 47, 47, 47, 47, 47, 47]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, H, W, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nerf_pytorch_paeng_amd import synthetic
    from nerf_pytorch_paeng_amd.dist import render_frame
    from oracle import restate as R

    sd = synthetic.make_state_dict(3, 4, 128)
    K, _, _ = synthetic.lego_camera()
    s = W / 800.0
    K = K.copy(); K[0, 0] *= s; K[1, 1] *= s; K[0, 2] = W / 2; K[1, 2] = H / 2
    pose = synthetic.pose_spherical(20.0, -30.0, 4.0)
    cfg = R.PathConfig(N_samples_c=16, N_samples_f=16, netDepth=4, netWidth=128)
    o, d = R.make_o_d(W, H, K, torch.from_numpy(pose[:3, :4]))

    def render_rows(r0, nr):     # this rank's rows, jitter keyed by GLOBAL ray index
        oo, dd = o[r0:r0 + nr].reshape(-1, 3), d[r0:r0 + nr].reshape(-1, 3)
        t = torch.from_numpy(R.counter_uniform(9, 0, r0 * W, nr * W, 16))
        u = torch.from_numpy(R.counter_uniform(9, 1, r0 * W, nr * W, 16))
        out = R.render_rays(torch.cat([oo, dd], -1), sd, cfg, t, u)
        return torch.cat([out["rgb_f"], out["disp_f"][:, None]], -1)

    rgb, disp = render_frame(H, W, K, pose, None, None, render_rows_fn=render_rows)
    np.save(os.path.join(out_dir, f"rgb_{rank}.npy"), rgb.numpy())
    np.save(os.path.join(out_dir, f"disp_{rank}.npy"), disp.numpy())
    if rank == 0:
        full = render_rows(0, H)
        np.save(os.path.join(out_dir, "ref.npy"), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("H,W", [(6, 5), (7, 4)])     # even and ragged row splits
def test_sharded_frame_equals_single_process(tmp_path, H, W):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), H, W, str(tmp_path)), nprocs=world, join=True)
    ref = np.load(tmp_path / "ref.npy")
    for r in range(world):
        rgb = np.load(tmp_path / f"rgb_{r}.npy")
        disp = np.load(tmp_path / f"disp_{r}.npy")
        assert rgb.shape == (H, W, 3) and disp.shape == (H, W)
        np.testing.assert_array_equal(rgb.reshape(-1, 3), ref[:, :3])      # bit-identical assembly on every rank
        np.testing.assert_array_equal(disp.reshape(-1), ref[:, 3])


def _tile_of_rows(r0, nr, W):
    """A stand-in renderer for plumbing tests: channel c of global ray g is a fixed function of (g, c)."""
    g = torch.arange(r0 * W, (r0 + nr) * W, dtype=torch.float32)
    return torch.stack([g * 0.25 + c for c in range(4)], -1)


def _worker8(rank, world, port, H, W, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nerf_pytorch_paeng_amd.dist import render_frame, shard_rows
        seen = []

        def rows(r0, nr):
            seen.append((r0, nr))
            return _tile_of_rows(r0, nr, W)

        rgb, disp = render_frame(H, W, None, None, None, None, render_rows_fn=rows)
        assert seen == [shard_rows(H, world, rank)]
        np.save(os.path.join(out_dir, f"frame_{rank}.npy"), torch.cat([rgb.reshape(-1, 3), disp.reshape(-1, 1)], -1).numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("H,W", [(800, 3), (378, 5)])       # config #3's 8 x 100 rows; fern's ragged 48, 48, 47 x 6 rows
def test_eight_rank_gather_assembles_the_frame(tmp_path, H, W):
    """The shard / pad / all_gather_into_tensor / un-pad path of dist.render_frame at the world size the driver's 8-GPU run uses
    (gloo, 8 CPU processes; the per-rank renderer is a stand-in): every rank ends with the whole frame, rows in order."""
    world = 8
    mp.spawn(_worker8, args=(world, _free_port(), H, W, str(tmp_path)), nprocs=world, join=True)
    ref = _tile_of_rows(0, H, W).numpy()
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / f"frame_{r}.npy"), ref)


def test_assemble_tiles_checks_the_split():
    from nerf_pytorch_paeng_amd.dist import assemble_tiles, shard_rows
    H, W = 378, 4
    tiles = [_tile_of_rows(*shard_rows(H, 8, r), W) for r in range(8)]
    assert torch.equal(assemble_tiles(tiles, H, W), _tile_of_rows(0, H, W))
    with pytest.raises(ValueError):
        assemble_tiles(tiles[:7] + [tiles[7][:-1]], H, W)


def test_c_abi_route_refuses_host_tensors_instead_of_falling_back():
    """gather_tiles(via="c_abi") is the library's RCCL route: HIP tensors only.  On host tensors (a gloo rehearsal without a GPU) it raises -- it
    does not quietly take the torch route -- and an unknown route name is an error."""
    from nerf_pytorch_paeng_amd import dist as mdist
    from nerf_pytorch_paeng_amd._lib import MiNerfError
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        tile = torch.rand(8 * 5, 4)
        assert torch.equal(mdist.gather_tiles(tile, 8, 5, force_collective=True, via="torch"), tile)
        with pytest.raises(MiNerfError, match="HIP device"):
            mdist.gather_tiles(tile, 8, 5, force_collective=True, via="c_abi")
        with pytest.raises(ValueError):
            mdist.gather_tiles(tile, 8, 5, via="mpi")
        assert mdist.gather_tiles(tile, 8, 5, via="c_abi") is tile          # a group of one without force_collective: nothing to gather on either route
    finally:
        dist.destroy_process_group()


def _from_group_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    if rank == 1:
        os.environ["MI_NERF_RCCL_LIB"] = "/nonexistent/librccl_of_rank_1.so"       # this rank alone cannot resolve RCCL
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nerf_pytorch_paeng_amd._lib import MiNerfError
        from nerf_pytorch_paeng_amd.dist import TileComm
        try:
            TileComm.from_group(torch.device("cuda", 0))
            verdict = "made a communicator"
        except MiNerfError as e:
            verdict = "MiNerfError: " + str(e)
        with open(os.path.join(out_dir, f"verdict_{rank}.txt"), "w") as fh:
            fh.write(verdict)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_from_group_raises_on_every_rank_when_one_rank_has_no_rccl(tmp_path):
    """TileComm.from_group is collective: a rank that cannot load librccl must not leave the others inside the id broadcast / ncclCommInitRank.
    The ranks agree first (all-reduce MIN of a flag); all of them raise MiNerfError, the one at fault names the loader's reason."""
    mp.spawn(_from_group_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    v0, v1 = (open(tmp_path / f"verdict_{r}.txt").read() for r in (0, 1))
    assert v0.startswith("MiNerfError") and "not usable on every rank" in v0
    assert v1.startswith("MiNerfError") and "librccl_of_rank_1" in v1
