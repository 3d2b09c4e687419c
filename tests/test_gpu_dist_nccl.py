"""Multi-GPU legs on hardware.

* `dist.render_frame` with backend "nccl" (= RCCL over xGMI) on 2 ranks, one GPU each: every rank's assembled frame must be
  bit-identical to the 1-rank frame (rows sharded, jitter keyed on the global ray index, ONE all-gather).  Needs two GPUs:
  skips itself on a one-GPU box (the driver's 8-GPU node runs it).
* `python bench.py --gpus 2` without a torchrun environment: the parent only launches the two workers (it makes no GPU call);
  rehearsed with BENCH_BACKEND=gloo so that both ranks can share the single GPU of the test box.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene(dev, H, W):
    from types import SimpleNamespace
    from nerf_pytorch_paeng_amd import synthetic, weights
    sd = synthetic.make_state_dict(5, 4, 128)
    packed = weights.PackedNeRF.from_state_dict(sd, dev)
    K, _, _ = synthetic.lego_camera()
    K = K.copy(); K[0, 0] *= W / 800.0; K[1, 1] *= W / 800.0; K[0, 2] = W / 2; K[1, 2] = H / 2
    pose = synthetic.pose_spherical(20.0, -30.0, 4.0)
    opts = SimpleNamespace(near=2.0, far=6.0, N_samples_c=16, N_samples_f=16, perturb=1.0, chunk_rays=4096, chunk_pts=524288,
                           data_type="blender", gpu_ids=[0], rank=0)
    return packed, K, pose, opts


def _nccl_worker(rank, world, port, H, W, out_dir, via="torch"):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from nerf_pytorch_paeng_amd.dist import render_frame
        packed, K, pose, opts = _scene(dev, H, W)
        rgb, disp = render_frame(H, W, K, pose, packed, opts, seed=3, via=via)
        torch.cuda.synchronize(dev)
        np.save(os.path.join(out_dir, f"rgb_{rank}.npy"), rgb.cpu().numpy())
        np.save(os.path.join(out_dir, f"disp_{rank}.npy"), disp.cpu().numpy())
        dist.barrier()
    finally:
        from nerf_pytorch_paeng_amd.dist import close_tile_comms
        close_tile_comms()
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("via", ["torch", "c_abi"])         # torch.distributed's all-gather | mi_nerf_all_gather_tiles of the C ABI
@pytest.mark.parametrize("H,W", [(24, 20), (25, 12)])      # even and ragged row splits
def test_nccl_two_rank_frame_is_bit_identical_to_one_rank(tmp_path, H, W, via):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL wants one GPU per rank); runs on the multi-GPU node")
    mp.spawn(_nccl_worker, args=(2, _free_port(), H, W, str(tmp_path), via), nprocs=2, join=True)
    from nerf_pytorch_paeng_amd.dist import render_frame
    dev = torch.device("cuda:0")
    packed, K, pose, opts = _scene(dev, H, W)
    rgb, disp = render_frame(H, W, K, pose, packed, opts, seed=3)        # no process group: one rank
    for r in range(2):
        np.testing.assert_array_equal(np.load(tmp_path / f"rgb_{r}.npy"), rgb.cpu().numpy())
        np.testing.assert_array_equal(np.load(tmp_path / f"disp_{r}.npy"), disp.cpu().numpy())


def _nccl_one_rank_worker(_index, port, H, W, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from nerf_pytorch_paeng_amd import dist as mdist
        packed, K, pose, opts = _scene(dev, H, W)
        rgb, disp = mdist.render_frame(H, W, K, pose, packed, opts, seed=3)
        tile = torch.cat([rgb.reshape(-1, 3), disp.reshape(-1, 1)], -1).contiguous()
        full = mdist.gather_tiles(tile, H, W, force_collective=True)             # all_gather_into_tensor on RCCL itself
        t = torch.tensor([1.5], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        torch.cuda.synchronize(dev)
        np.save(os.path.join(out_dir, "tile.npy"), tile.cpu().numpy())
        np.save(os.path.join(out_dir, "full.npy"), full.cpu().numpy())
        assert float(t.item()) == 1.5
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_rccl_call_path_with_one_rank(tmp_path):
    """What a one-GPU box can exercise of the RCCL leg: init_process_group("nccl", device_id=...), the tile all-gather, the timing
    all-reduce and the barrier run against RCCL in a group of one rank (own process: the group must not leak into other tests)."""
    H, W = 24, 20
    mp.spawn(_nccl_one_rank_worker, args=(_free_port(), H, W, str(tmp_path)), nprocs=1, join=True)
    np.testing.assert_array_equal(np.load(tmp_path / "tile.npy"), np.load(tmp_path / "full.npy"))


def _c_abi_one_rank_worker(_index, port, out_dir):
    """The C ABI's own communicator in a group of one rank: (a) bootstrapped over torch.distributed and used through gather_tiles(via="c_abi"),
    (b) stand-alone from a unique id with no process group at all, (c) in place (the tile already lies in the frame), (d) on a side stream."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from nerf_pytorch_paeng_amd import dist as mdist
    res = {}
    gen = torch.Generator(device=dev).manual_seed(0)
    # (b) first, before any process group exists: the helpers need nothing from torch.distributed
    comm = mdist.TileComm(mdist.TileComm.unique_id(), 1, 0, dev)
    for name, (H, W) in (("lego", (800, 800)), ("fern", (378, 504))):
        tile = torch.rand(H * W, 4, generator=gen, device=dev)
        res[f"alone_{name}"] = bool(torch.equal(comm.all_gather_tiles(tile, H, W), tile))
    frame = torch.rand(800 * 800, 4, generator=gen, device=dev)                  # (c) sendbuff == recvbuff: RCCL's in-place form
    want = frame.clone()
    res["in_place"] = bool(torch.equal(comm.all_gather_tiles(frame, 800, 800, out=frame), want)) and frame.data_ptr() == frame.data_ptr()
    side = torch.cuda.Stream(dev)                                                 # (d) ordered on the stream the producer ran on
    with torch.cuda.stream(side):
        tile = torch.rand(378 * 504, 4, generator=gen, device=dev) * 2.0 + 1.0
        out = comm.all_gather_tiles(tile, 378, 504)
    side.synchronize()
    res["side_stream"] = bool(torch.equal(out, tile))
    try:
        comm.all_gather_tiles(tile[:-504], 378, 504)
        res["bad_rows_refused"] = False
    except mdist.MiNerfError as e:
        res["bad_rows_refused"] = "owns 378 of 378 rows" in str(e)
    comm.close()
    try:
        comm.all_gather_tiles(tile, 378, 504)
        res["closed_refused"] = False
    except mdist.MiNerfError:
        res["closed_refused"] = True
    # (a) the route dist.render_frame / bench.py take
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        for name, (H, W) in (("lego", (800, 800)), ("fern", (378, 504))):
            tile = torch.rand(H * W, 4, generator=gen, device=dev)
            a = mdist.gather_tiles(tile, H, W, force_collective=True, via="c_abi")
            b = mdist.gather_tiles(tile, H, W, force_collective=True, via="torch")
            torch.cuda.synchronize(dev)
            res[f"group_{name}"] = bool(torch.equal(a, b) and torch.equal(a, tile))
        mdist.close_tile_comms()
    finally:
        dist.destroy_process_group()
    with open(os.path.join(out_dir, "res.json"), "w") as f:
        json.dump(res, f)


@pytest.mark.timeout(600)
def test_c_abi_tile_gather_with_one_rank(tmp_path):
    """mi_nerf_comm_unique_id / _init_rank / mi_nerf_all_gather_tiles / _destroy against RCCL itself (librccl resolved at first use), in a
    group of ONE rank -- all a one-GPU box can run: bit-equal to torch.distributed's all-gather for the 800-row and the 378-row frame."""
    mp.spawn(_c_abi_one_rank_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    res = json.load(open(tmp_path / "res.json"))
    assert res and all(res.values()), res
    assert set(res) == {"alone_lego", "alone_fern", "in_place", "side_stream", "bad_rows_refused", "closed_refused", "group_lego", "group_fern"}


def _fake_rccl_worker(rank, world, port, cases, out_dir):
    """One of `world` processes SHARING cuda:0: the C ABI's gather with tests/c_abi/fake_rccl.cpp in librccl's place (MI_NERF_RCCL_LIB, set by the
    parent), the unique id travelling over a gloo group."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    try:
        from nerf_pytorch_paeng_amd import dist as mdist
        comm = mdist.TileComm.from_group(dev)
        assert (comm.world, comm.rank) == (world, rank)
        for H, W, C in cases:
            frame = torch.rand(H * W, C, generator=torch.Generator().manual_seed(H * 131 + W + C)).to(dev)     # the same on every rank
            r0, nr = mdist.shard_rows(H, world, rank)
            tile = frame[r0 * W:(r0 + nr) * W].clone()
            got = comm.all_gather_tiles(tile, H, W)
            res[f"{H}x{W}x{C}"] = bool(torch.equal(got, frame))
            if H % world == 0:                                    # in place: the tile already lies in the frame at its own block
                buf = torch.full((H * W, C), float("nan"), device=dev)
                buf[r0 * W:(r0 + nr) * W] = tile
                comm.all_gather_tiles(buf[r0 * W:(r0 + nr) * W], H, W, out=buf)
                res[f"{H}x{W}x{C}_in_place"] = bool(torch.equal(buf, frame))
            try:                                                  # a tile of the neighbour's size is refused where the blocks differ
                other = mdist.shard_rows(H, world, (rank + 1) % world)[1]
                if other != nr:
                    comm.all_gather_tiles(frame[:other * W].clone(), H, W)
                    res[f"{H}x{W}x{C}_wrong_rows_refused"] = False
            except mdist.MiNerfError:
                res[f"{H}x{W}x{C}_wrong_rows_refused"] = True
            dist.barrier()
        # the route dist.render_frame / bench.py take, on a rendered frame
        packed, K, pose, opts = _scene(dev, 25, 12)
        rgb, disp = mdist.render_frame(25, 12, K, pose, packed, opts, seed=3, via="c_abi")
        np.save(os.path.join(out_dir, f"rgb_{rank}.npy"), rgb.cpu().numpy())
        comm.close()
        mdist.close_tile_comms()
    finally:
        dist.destroy_process_group()
    with open(os.path.join(out_dir, f"res_{rank}.json"), "w") as f:
        json.dump(res, f)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 3, 4])
def test_c_abi_tile_gather_between_ranks_with_a_stand_in_for_rccl(tmp_path, world, fake_rccl_lib):
    """mi_nerf_all_gather_tiles with MORE THAN ONE rank on the one-GPU box: RCCL refuses two ranks on a device, so tests/c_abi/fake_rccl.cpp (an
    all-gather between processes through host shared memory, same entry points) is loaded in its place via MI_NERF_RCCL_LIB.  What runs for real
    is everything of the C ABI around the collective: the communicator's rank / world plumbing, equal blocks straight into the frame and in place,
    ragged splits (378 rows over 2 / 3 / 4 ranks, 4096 rays over 3) padded in the staging buffer, gathered in place there and un-padded by the copy
    kernel -- every rank's frame must equal the whole, bit for bit -- and dist.render_frame(via="c_abi") against the one-rank frame."""
    lib = fake_rccl_lib
    cases = [(378, 504, 4), (800, 800, 4), (4096, 1, 4), (25, 3, 1), (24, 20, 4)]
    old = os.environ.get("MI_NERF_RCCL_LIB")
    os.environ["MI_NERF_RCCL_LIB"] = lib                          # inherited by the spawned ranks; this process never resolves RCCL itself
    try:
        mp.spawn(_fake_rccl_worker, args=(world, _free_port(), cases, str(tmp_path)), nprocs=world, join=True)
    finally:
        if old is None:
            os.environ.pop("MI_NERF_RCCL_LIB", None)
        else:
            os.environ["MI_NERF_RCCL_LIB"] = old
    from nerf_pytorch_paeng_amd.dist import render_frame
    dev = torch.device("cuda:0")
    packed, K, pose, opts = _scene(dev, 25, 12)
    rgb, _ = render_frame(25, 12, K, pose, packed, opts, seed=3)           # no process group: one rank
    for rank in range(world):
        res = json.load(open(tmp_path / f"res_{rank}.json"))
        assert res and all(res.values()), (rank, res)
        assert any(k.endswith("_in_place") for k in res) or world == 3 and not any(H % 3 == 0 for H, _, _ in cases)
        np.testing.assert_array_equal(np.load(tmp_path / f"rgb_{rank}.npy"), rgb.cpu().numpy())


@pytest.mark.timeout(600)
def test_bench_line_survives_a_wedged_c_abi_collective(fake_rccl_lib):
    """The `collective.c_abi` leg runs on a watched thread and a HIP stream of its own: if one of its collectives never returns (here: the stand-in's 4th
    all-gather sleeps for ever on every rank) the ranks give up after BENCH_C_ABI_TIMEOUT_S, issue no further GPU or process-group call, rank 0 still prints
    the ONE line -- with `c_abi: {"error": "hung..."}` and everything the torch route measured -- and the job ends with status 75: a hang is not a success."""
    lib = fake_rccl_lib
    env = dict(os.environ, BENCH_BACKEND="gloo", MI_NERF_RCCL_LIB=lib, FAKE_RCCL_HANG_AFTER="3", BENCH_C_ABI_TIMEOUT_S="5", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames", "1", "--no-cpu-baseline", "--no-bf16-leg",
                        "--no-f16s-leg", "--scaling", "strong"], env=env, capture_output=True, text=True, timeout=500)
    assert r.returncode == 75, (r.returncode, r.stderr[-3000:])
    assert "the C-ABI gather leg hung" in r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    c = line["collective"]
    assert c["c_abi"]["error"].startswith("hung") and c["frame_equal_across_ranks"] is True and c["all_gather_ms"] > 0 and c["tile_gather_route"] == "torch"
    assert line["frame_ms"] > 0 and line["frame_ms_c_abi"] is None and line["frame_checksum_c_abi"] is None


@pytest.mark.parametrize("world,H,W,C", [(8, 378, 504, 4), (8, 800, 800, 4), (3, 4096, 1, 4), (4, 25, 3, 1), (6, 800, 800, 3), (5, 7, 1, 1)])
def test_unpad_tiles_for_any_world_size(world, H, W, C):
    """The ragged half of mi_nerf_all_gather_tiles without RCCL: a staging buffer as the padded in-place all-gather leaves it for `world`
    ranks (NaN in the padding) -> the frame, bit for bit; float4 and scalar forms."""
    from nerf_pytorch_paeng_amd import dist as mdist
    dev = torch.device("cuda:0")
    frame = torch.rand(H * W, C, generator=torch.Generator(device=dev).manual_seed(world * H), device=dev)
    max_rows = H // world + (1 if H % world else 0)
    staging = torch.full((world, max_rows * W * C), float("nan"), device=dev)
    for r in range(world):
        r0, nr = mdist.shard_rows(H, world, r)
        staging[r, :nr * W * C] = frame[r0 * W:(r0 + nr) * W].reshape(-1)
    got = mdist.unpad_tiles(staging, world, H, W, C)
    assert torch.equal(got, frame)
    assert int(_lib_staging_bytes(world, H, W, C)) == (0 if H % world == 0 else staging.numel() * 4)


def _lib_staging_bytes(world, H, W, C):
    from nerf_pytorch_paeng_amd._lib import lib
    return lib().mi_nerf_all_gather_staging_bytes(world, H, W, C)


@pytest.mark.timeout(900)
def test_bench_both_gather_routes_through_rccl_with_one_rank():
    """`python bench.py` with BENCH_FORCE_DIST=1: the worker's RCCL set-up, barriers and max-over-ranks all-reduce at world size 1, and the `collective`
    object of an N > 1 line against RCCL ITSELF (what a one-GPU box can show): backend "nccl", the device all-gathered, the tile all-gather timed by
    hipEvents, the frame checksum and the re-rendered block equal -- and the same frames timed through BOTH routes in the one run: torch.distributed
    (frame_ms) and the library's own communicator on its side stream (frame_ms_c_abi), with equal checksums."""
    env = dict(os.environ, BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--frames", "1", "--train-steps", "0",
                        "--no-cpu-baseline", "--no-small-batch", "--no-bf16-leg", "--no-f16s-leg"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 1e5
    c = line["collective"]
    assert c["backend"] == "nccl" and c["world_size"] == 1 and c["distinct_devices"] == 1 and c["ranks"][0]["cus"] == 256
    assert c["frame_equal_across_ranks"] is True and c["neighbour_tile_recomputed_equal"] is True
    assert 0 < c["all_gather_ms"] < 50 and c["all_gather_bytes_assembled"] == 800 * 800 * 16
    # the same tile through the C ABI's own communicator (mi_nerf_all_gather_tiles), timed beside it and bit-equal ...
    assert c["tile_gather_route"] == "torch" and "error" not in c["c_abi"], c["c_abi"]
    assert c["c_abi"]["equal_to_torch_route_on_every_rank"] is True and 0 < c["c_abi"]["all_gather_ms"] < 50 and c["c_abi"]["world_size"] == 1
    # ... and as the route of the timed frames themselves
    assert c["c_abi"]["frame_checksum_equals_torch_route_on_every_rank"] is True and c["c_abi"]["frames"] == 1
    assert line["frame_ms_c_abi"] > 0 and line["frame_checksum_c_abi"] == line["frame_checksum"] == c["frame_checksum_rank0"]
    assert 0.5 < line["frame_ms_c_abi"] / line["frame_ms"] < 2.0


@pytest.mark.timeout(900)
def test_bench_under_torch_distributed_run():
    """The driver's own form for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P
    bench.py --gpus 2 ...` -- the workers are used as launched (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), rank 0
    prints ONE line, and the line carries the self-verifying `collective` block.  gloo when the box has one GPU (the ranks share it)."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if torch.cuda.device_count() < 2:
        env["BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--frames", "1",
                        "--no-cpu-baseline", "--no-bf16-leg", "--no-f16s-leg"], env=env, capture_output=True, text=True, timeout=850, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["scaling"] == "strong" and j["config"]["rays_per_gpu"] == 2048
    c = j["collective"]
    assert c["world_size"] == 2 and [x["rank"] for x in c["ranks"]] == [0, 1]
    assert c["frame_equal_across_ranks"] is True and c["neighbour_tile_recomputed_equal"] is True
