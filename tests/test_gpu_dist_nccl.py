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


def _nccl_worker(rank, world, port, H, W, out_dir):
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
        rgb, disp = render_frame(H, W, K, pose, packed, opts, seed=3)
        torch.cuda.synchronize(dev)
        np.save(os.path.join(out_dir, f"rgb_{rank}.npy"), rgb.cpu().numpy())
        np.save(os.path.join(out_dir, f"disp_{rank}.npy"), disp.cpu().numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("H,W", [(24, 20), (25, 12)])      # even and ragged row splits
def test_nccl_two_rank_frame_is_bit_identical_to_one_rank(tmp_path, H, W):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL wants one GPU per rank); runs on the multi-GPU node")
    mp.spawn(_nccl_worker, args=(2, _free_port(), H, W, str(tmp_path)), nprocs=2, join=True)
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


@pytest.mark.timeout(900)
def test_bench_one_rank_through_rccl():
    """`python bench.py` with BENCH_FORCE_DIST=1: the worker's RCCL set-up, barriers and max-over-ranks all-reduce at world size 1."""
    env = dict(os.environ, BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--frames", "0", "--train-steps", "0",
                        "--no-cpu-baseline", "--no-small-batch", "--no-bf16-leg"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 1e5
    assert "collective" not in line                       # --frames 0: nothing to gather


@pytest.mark.timeout(900)
def test_bench_collective_block_through_rccl_with_one_rank():
    """The `collective` object of an N > 1 line against RCCL itself, in a group of one rank (what a one-GPU box can show): backend "nccl",
    the device all-gathered, the tile all-gather timed by hipEvents, the frame checksum and the re-rendered block equal."""
    env = dict(os.environ, BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--frames", "1", "--train-steps", "0",
                        "--no-cpu-baseline", "--no-small-batch", "--no-bf16-leg", "--no-f16s-leg"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    c = line["collective"]
    assert c["backend"] == "nccl" and c["world_size"] == 1 and c["distinct_devices"] == 1 and c["ranks"][0]["cus"] == 256
    assert c["frame_equal_across_ranks"] is True and c["neighbour_tile_recomputed_equal"] is True
    assert 0 < c["all_gather_ms"] < 50 and c["all_gather_bytes_assembled"] == 800 * 800 * 16


@pytest.mark.timeout(900)
def test_bench_launches_its_own_workers():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if torch.cuda.device_count() < 2:
        env["BENCH_BACKEND"] = "gloo"                   # two ranks on the one GPU: plumbing rehearsal
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--frames", "1",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints ONE line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["scaling"] == "strong"
    assert j["config"]["rays_per_gpu"] == 2048 and j["value"] > 0 and j["value_weak"] > 0
    assert j["frame_ms_800x800"] > 0 and j["roofline"]["frac"] <= 1.0


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
