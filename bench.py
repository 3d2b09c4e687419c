#!/usr/bin/env python3
"""Benchmark of the NeRF volume-rendering hot path on MI355X (driver contract: one JSON line on rank 0).

    python bench.py --gpus N --steps K --warmup W          # N > 1: this process only LAUNCHES N workers (one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             # or let torchrun start the workers

Metric (BASELINE.json): rays/sec on a 4096-ray batch with 64 coarse + 128 fine samples through the 8x256
coarse/fine MLPs, fp32 -- BASELINE config #2, "lego coarse+fine 4096 rays, 64+128 samples, 8x256 MLP,
1xMI355X fp32".  One step = one full render_rays pass (stratified sampling with its jitter draw, nerf_process.py:58-60 -> coarse
MLP -> composite -> inverse-CDF resampling with its jitter draw, :162-163, + merge -> fine MLP over all 192
depths -> composite) over one batch whose rays are resident in HBM; the jitter is drawn INSIDE the step, by the
sampling kernels themselves (counter-based generator keyed on the global ray index).

Timing: W untimed warm-up steps, then EXACTLY K timed steps between barrier + torch.cuda.synchronize() on both sides, max over ranks.  In front
of the W warm-up steps the same step runs for 60 ms (BENCH_PREWARM_S; untimed, like them; `config.prewarm_ms`): the clock governor takes tens of
milliseconds to come up from idle and five steps of a 512-ray shard are 5 ms (profiles/r04_prewarm_512_ray_shard.txt); the 4096-ray step is unaffected.

`ms_per_step_median` (SURVEY.md 8(d)'s protocol, beside the contract's mean): max(K, 3) FURTHER steps, each bracketed by hipEvents on the launch stream, median, max
over ranks; `value` does not come from it.

Scaling: the 4096-ray batch is SHARDED over the N GPUs -- 4096/N contiguous rays per rank
(512 at N = 8), no data-path collective -- so `value` = 4096 * K / max-over-ranks time is STRONG scaled.
`value_weak` (every rank renders its own 4096-ray batch, N * 4096 * K / time) is measured in a second leg of
the same run for N > 1; at N = 1 the two are the same measurement.

Also on the line: the 800x800 frame time (rows sharded over the ranks, ONE all-gather of the output tiles
over RCCL), `roofline` for the dominant kernel (the fine-network fused MLP launch, hipEvent-timed on its
launch stream), `bf16` (BASELINE config #5: the same step on the bf16 MFMA variant with its PSNR against the
fp32 outputs of the same rays, and its own `small_batch`), `f16_split` (the split-precision variant: f16 hi + lo
operands on the f16 matrix pipe, fp32-grade results; an extra leg, never `value`; `train.f16_split` is the training step with its
three MFMA kernels in that mode), `small_batch` (the same step at 256..2048 rays on
one GPU: what a rank sees under strong scaling) and `cpu_baseline` (the CPU oracle timed on the host cores;
rank 0, N = 1 only).

N > 1 lines carry `collective` (collective_block): what the process group saw, the tile all-gather timed, the frame's checksum equal on every rank, a neighbour's tile
re-rendered and compared -- and, on backend "nccl", `collective.c_abi` (c_abi_route_leg): the same tile gathered through the C ABI's own RCCL communicator
(mi_nerf_all_gather_tiles), timed and compared bit for bit with the torch.distributed route, and then THE SAME --frames POSES rendered and assembled through that
route: `frame_ms_c_abi` / `frame_checksum_c_abi` beside `frame_ms` / `frame_checksum`, so ONE `--gpus N` run times both routes.  That leg runs on a watched thread
and a HIP stream of its own.  Exit status: 75 when one of its collectives never returned (the line is printed first, with `c_abi.error`), 76 when backend "nccl" ran
but `collective.distinct_devices` != N (RCCL did not see N GPUs); 0 otherwise.

BENCH_SOLO_RANK=1 with RANK / WORLD_SIZE set rehearses ONE rank's share of an N-rank run alone (no process
group; the frame leg renders this rank's rows only): the GPU pool admits at most 6 processes on a card, so an
8-rank run cannot be rehearsed on a one-GPU box with 8 live ranks.

Synthetic inputs: lego camera geometry, pose_spherical(0,-30,4), 4096 pixels from
RandomState(0), Xavier(seed 0) weights with the density head x20, counter-based jitter seed 0.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic work: 2 x MACs of the Linear layers only
FLOP_PER_POINT = 2 * 593408              # 8x256 network, model/NeRF.py:24-30
N_RAYS, SC, NF = 4096, 64, 128
POINTS_PER_RAY = SC + (SC + NF)          # 64 coarse + 192 fine network evaluations
FLOP_PER_RAY = FLOP_PER_POINT * POINTS_PER_RAY          # 303,824,896
PEAK_F32_MFMA_TFLOPS = 157.3             # MI355X_MICROARCH.md: 256 CU x 256 FLOP/clk x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0           # dense bf16 MFMA peak (the headline 5 PF figure includes 2:1 sparsity)
LDS_DMA_CHIP_TBPS = 34.8                 # chip-wide L2 -> LDS fill rate MEASURED with the kernel's own geometry, 4 waves per CU issuing (tools/mfma_probe6 "dma",
                                         # profiles/r03_bf16_hybrid_probe.txt); the 6.4 TB/s of MI355X_MICROARCH.md is for one loader wave per CU
BF16_POINTS_PER_PASS = 256               # mlp_bf16.hip: a workgroup takes 4 waves x 64 points through one pass of the weight stream
PREWARM_S = float(os.environ.get("BENCH_PREWARM_S", "0.06"))                 # untimed steps in front of the W warm-up steps: clock ramp from idle (timed_steps)
LAUNCHER_GRACE_S = float(os.environ.get("BENCH_LAUNCHER_GRACE_S", "10"))     # SIGTERM -> SIGKILL for the survivors of a failed run
KERNEL_SOURCES = ("mlp_fp32.hip", "mlp_core.h", "layout.h", "common.h")
METRIC_FALLBACK = "rays/sec (4096-ray batch, 64c+128f samples) + 800\u00d7800 frame render ms"


def baseline_metric() -> str:
    """The metric string, byte for byte as BASELINE.json spells it (800 U+00D7 800); the literal only when the file did not travel."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json"), encoding="utf-8") as fh:
            m = json.load(fh)["metric"]
        return m if isinstance(m, str) and m else METRIC_FALLBACK
    except (OSError, ValueError, KeyError):
        return METRIC_FALLBACK


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=2, help="timed 800x800 frames (0 disables the frame metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-steps", type=int, default=None,
                    help="timed training steps (forward + hand-written backward + Adam, train.py:53-70) reported as `train`; "
                         "default 8 on one GPU, 0 on several")
    ap.add_argument("--bf16", action="store_true", help="bf16 MFMA variant (BASELINE config #5)")
    ap.add_argument("--workload", choices=["lego", "fern"], default="lego",
                    help="lego: BASELINE config #2 (default, the headline metric); fern: config #4, LLFF geometry + NDC rays")
    ap.add_argument("--scaling", choices=["strong", "weak", "both"], default="both",
                    help="strong: the 4096-ray batch sharded over the GPUs (value); weak: 4096 rays per GPU (value_weak); both (default)")
    ap.add_argument("--no-small-batch", action="store_true", help="skip the 256..2048-ray legs (N = 1)")
    ap.add_argument("--no-bf16-leg", action="store_true", help="skip the bf16 leg (BASELINE config #5) of the default fp32 run")
    ap.add_argument("--no-f16s-leg", action="store_true", help="skip the split-precision (f16 hi + lo operands, fp32-grade results) leg of the default fp32 run")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without a torchrun environment.  This process makes NO GPU call (it does
# not even import torch); it starts one worker per GPU as a child process and waits for them.
# ----------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_workers(n: int) -> int:
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC (RCCL across processes); a value the user set wins
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env))
    rc = 0
    alive = set(range(n))
    kill_at = None                                # deadline for the survivors of a failed run (SIGTERM sent, SIGKILL next)
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                sys.stderr.write(f"bench.py: worker rank {r} exited with status {code}; stopping the others\n")
                for o in alive:
                    procs[o].terminate()          # exact children of this process, by PID
                kill_at = time.monotonic() + LAUNCHER_GRACE_S
        if alive and kill_at is not None and time.monotonic() >= kill_at:
            # a rank blocked inside a collective or a driver call can sit out SIGTERM: do not wait for it forever
            for o in sorted(alive):
                sys.stderr.write(f"bench.py: worker rank {o} ignored SIGTERM for {LAUNCHER_GRACE_S:.0f} s; killing it\n")
                procs[o].kill()
            for o in sorted(alive):
                procs[o].wait()
            alive.clear()
        if alive:
            time.sleep(0.05)
    return rc


def kernel_build_id() -> str:
    """sha256 (first 16 hex digits) of the fused-MLP kernel's sources: ties a committed PMC figure to a build."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "nerf_pytorch_paeng_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def bf16_stream(blob, n_points: int, kernel_ms: float) -> dict:
    """The bf16 kernel's weight stream: every workgroup pulls the whole bf16 stream from L2 into LDS once per BF16_POINTS_PER_PASS
    points (the register file holds no more), so a launch moves passes x stream bytes through the L2 -> LDS path.  Reported beside
    the path's measured capacity: the kernel uses a fifth of it (round 2 took the 6.4 TB/s of one loader wave per CU for a ceiling)."""
    import numpy as np
    hdr = blob[:64].cpu().numpy().view(np.uint32)
    stream_bytes = int(hdr[8])
    passes = (n_points + BF16_POINTS_PER_PASS - 1) // BF16_POINTS_PER_PASS
    gb = passes * stream_bytes / 1e9
    tbps = gb / kernel_ms
    return {"bytes_per_pass": stream_bytes, "passes": passes, "GB_per_launch": round(gb, 3), "l2_to_lds_TBps": round(tbps, 2),
            "chip_lds_dma_rate_TBps_measured": LDS_DMA_CHIP_TBPS, "frac_of_lds_dma_rate": round(tbps / LDS_DMA_CHIP_TBPS, 3)}


def frame_checksum(torch, rgb, disp) -> int:
    """Order-sensitive 64-bit checksum over the fp32 BIT PATTERNS of an assembled frame (rgb [H,W,3], disp [H,W]): equal checksums on two
    lines -- an N = 1 run and an N = 8 run of the same command -- mean the frames are bit-identical whatever the number of GPUs."""
    t = torch.cat([rgb.reshape(-1, 3), disp.reshape(-1, 1)], -1)
    b = t.contiguous().view(torch.int32).to(torch.int64).reshape(-1)
    return int((b * (torch.arange(b.numel(), device=b.device, dtype=torch.int64) % 1000003 + 1)).sum().item())


C_ABI_LEG_TIMEOUT_S = float(os.environ.get("BENCH_C_ABI_TIMEOUT_S", "180"))
EXIT_C_ABI_HUNG = 75                 # a collective of the library's own RCCL communicator never returned on this rank (the line is still printed first)
EXIT_RCCL_SAW_FEWER_GPUS = 76        # backend "nccl" but the ranks did not sit on N distinct GPUs: the run says nothing about N GPUs
_c_abi_leg_hung = False


def exit_status(hung: bool, backend: str, world: int, collective) -> int:
    """What a worker leaves with once its line is out: EXIT_C_ABI_HUNG when the C-ABI gather leg never came back on this rank, EXIT_RCCL_SAW_FEWER_GPUS
    when backend "nccl" ran with `world` ranks on fewer than `world` distinct devices (host + PCI bus id), else 0."""
    if hung:
        return EXIT_C_ABI_HUNG
    if collective is not None and backend == "nccl" and collective.get("distinct_devices") != world:
        return EXIT_RCCL_SAW_FEWER_GPUS
    return 0


def c_abi_route_leg(dist, mdist, torch, dev, cdev, local, full_torch, H, W, n_frames, render_frame_c_abi, frame_pose, checksum_torch) -> dict:
    """The library's OWN route to the tile all-gather (mi_nerf_comm_* / mi_nerf_all_gather_tiles: csrc/comm.hip) measured next to the torch.distributed
    route, on every rank, in ONE run: (1) the communicator set-up (TileComm.from_group: the ranks agree that RCCL is loadable before anything collective);
    (2) the same tile gathered 12 times, hipEvent-timed, and compared bit for bit with the frame the torch route assembled; (3) when that holds on every
    rank, the same `--frames` poses rendered and assembled through this route between two barriers -> frame_ms_c_abi + frame_checksum_c_abi.

    The leg runs on a thread of its own AND a HIP stream of its own: if one of its collectives never returns, the stuck work sits on that side stream, the
    main thread stops waiting after C_ABI_LEG_TIMEOUT_S, issues NO further GPU or process-group call, prints the line with `c_abi.error` and leaves with
    EXIT_C_ABI_HUNG -- non-zero, so the launcher stops the other ranks and CI sees the hang."""
    import statistics
    import threading
    out = {}
    torch.cuda.synchronize(dev)                            # nothing of the main stream is pending: the side stream depends on no event of it
    side = torch.cuda.Stream(dev)

    def leg():
        try:
            torch.cuda.set_device(dev)
            with torch.cuda.stream(side):
                comm = mdist.tile_comm(dev)                # collective; raises MiNerfError on EVERY rank when any rank cannot load librccl
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                times = []
                for i in range(12):
                    side.synchronize()
                    dist.barrier()
                    ev[0].record()
                    full = comm.all_gather_tiles(local, H, W)
                    ev[1].record()
                    side.synchronize()
                    if i >= 2:
                        times.append(ev[0].elapsed_time(ev[1]))
                tm = torch.tensor([statistics.median(times)], dtype=torch.float64, device=cdev)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                eq = torch.tensor([int(torch.equal(full, full_torch))], dtype=torch.int32, device=cdev)
                dist.all_reduce(eq, op=dist.ReduceOp.MIN)
                out.update(all_gather_ms=round(float(tm.item()), 4), equal_to_torch_route_on_every_rank=bool(eq.item()), world_size=comm.world,
                           staging_bytes=int(mdist.lib().mi_nerf_all_gather_staging_bytes(comm.world, H, W, int(local.shape[1]))),
                           stream="a HIP stream of the leg's own (not torch's default stream): renders, gathers and hipEvents of this leg all sit on it",
                           what="mi_nerf_all_gather_tiles: libmi_nerf.so's own RCCL communicator (unique id broadcast over the process group), "
                                "enqueued on the stream the tile was rendered on; hipEvents on that stream, median of 10, max over ranks")
                if bool(eq.item()) and n_frames > 0:       # the timed frames of the line, once more, assembled by this route
                    render_frame_c_abi(frame_pose(-1))
                    side.synchronize()
                    dist.barrier()
                    t0 = time.perf_counter()
                    for f in range(n_frames):
                        rgb, disp = render_frame_c_abi(frame_pose(f))
                    side.synchronize()
                    dist.barrier()
                    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=cdev)
                    dist.all_reduce(el, op=dist.ReduceOp.MAX)
                    cs = frame_checksum(torch, rgb, disp)
                    same = torch.tensor([int(cs == checksum_torch)], dtype=torch.int32, device=cdev)
                    dist.all_reduce(same, op=dist.ReduceOp.MIN)
                    out.update(frame_ms=round(1e3 * float(el.item()) / n_frames, 2), frames=n_frames, frame_checksum=cs,
                               frame_checksum_equals_torch_route_on_every_rank=bool(same.item()))
                mdist.close_tile_comms()
        except Exception as e:                             # noqa: BLE001
            out["error"] = repr(e)

    t = threading.Thread(target=leg, daemon=True)
    t.start()
    t.join(C_ABI_LEG_TIMEOUT_S)
    if t.is_alive():
        global _c_abi_leg_hung
        _c_abi_leg_hung = True
        return {**{k: v for k, v in out.items() if k != "error"},
                "error": f"hung: the leg did not finish within {C_ABI_LEG_TIMEOUT_S:.0f} s (a collective of the library's RCCL communicator never returned); "
                         f"this rank leaves with exit status {EXIT_C_ABI_HUNG} after printing the line"}
    return out


def collective_block(dist, mdist, torch, dev, backend, rank, world, H, W, K, pose, packed, opts, bf16, rgb_last, disp_last) -> dict:
    """The `collective` object of an N > 1 line (every rank calls this; the dict is the same on all of them).

    backend / world_size        what torch.distributed itself reports (backend "nccl" IS RCCL on ROCm)
    ranks                       per rank: host, device ordinal, device name, PCI bus id -- all-gathered, so N distinct GPUs are visible in the line
    distinct_devices            number of distinct (host, PCI bus id) pairs among them (N on a real run; 1 in a gloo rehearsal on one GPU)
    all_gather_ms               ONE gather_tiles() of this frame's [rows_local * W, 4] tile: hipEvents on the stream the collective is
                                ordered on (torch's current stream), median of 10 after 2 warm-ups, max over ranks; + the bytes it moves
    frame_equal_across_ranks    the assembled frame of the last timed pose has the same 64-bit checksum on every rank (all-reduce MIN of
                                an equality flag against rank 0's checksum)
    neighbour_tile_recomputed_equal   every rank re-renders the row block of rank (r + 1) % N on ITS OWN GPU and compares it bit for bit
                                with that block of the gathered frame: the gather put each tile where it belongs, and the frame does
                                not depend on which GPU rendered which rows (dist.py's bit-identity claim, checked on the hardware)
    tile_gather_route           "torch": frame_ms / frame_checksum of the line are frames assembled by torch.distributed all_gather_into_tensor
    c_abi                       (filled in by the caller from c_abi_route_leg; backend "nccl", or a stand-in named by MI_NERF_RCCL_LIB) the same tile gathered
                                by the C ABI's route (include/mi_nerf.h, csrc/comm.hip), timed like all_gather_ms and compared bit for bit with the torch
                                route's frame on every rank, then the SAME --frames poses rendered + assembled through it (frame_ms / frame_checksum; the line
                                repeats them at top level as frame_ms_c_abi / frame_checksum_c_abi); {"error": ...} when RCCL cannot be loaded or the leg hangs

    Returns (block, local_tile, gathered_frame): the last two feed c_abi_route_leg."""
    import statistics
    cdev = dev if backend == "nccl" else torch.device("cpu")
    me = {"rank": rank, "host": "?", "device": f"cuda:{dev.index}", "name": "?", "pci_bus_id": None, "cus": 0}
    try:                                                   # rank-local look-ups must not be able to stop a rank short of the collectives below
        props = torch.cuda.get_device_properties(dev)
        bus = getattr(props, "pci_bus_id", None)
        me.update(host=socket.gethostname(), name=str(props.name), pci_bus_id=None if bus is None else int(bus), cus=int(props.multi_processor_count))
    except Exception as e:                                 # noqa: BLE001
        me["error"] = repr(e)
    ranks = [None] * world
    dist.all_gather_object(ranks, me)
    pose_last = pose
    local = mdist.render_shard(H, W, K, pose_last, packed, opts, world, rank, seed=0, bf16=bf16)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    times = []
    for i in range(12):
        torch.cuda.synchronize(dev)
        dist.barrier()
        ev[0].record()
        full = mdist.gather_tiles(local, H, W, force_collective=True)
        ev[1].record()
        torch.cuda.synchronize(dev)
        if i >= 2:
            times.append(ev[0].elapsed_time(ev[1]))
    tm = torch.tensor([statistics.median(times)], dtype=torch.float64, device=cdev)
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)

    cs = torch.tensor([frame_checksum(torch, rgb_last, disp_last)], dtype=torch.int64, device=cdev)
    cs0 = cs.clone()
    dist.broadcast(cs0, src=0)
    nb = (rank + 1) % world
    r0, nr = mdist.shard_rows(H, world, nb)
    mine = mdist.render_shard(H, W, K, pose_last, packed, opts, world, nb, seed=0, bf16=bf16)
    flags = torch.tensor([int(cs.item() == cs0.item()), int(torch.equal(mine, full[r0 * W:(r0 + nr) * W]))], dtype=torch.int32, device=cdev)
    dist.all_reduce(flags, op=dist.ReduceOp.MIN)
    max_rows = (H + world - 1) // world
    block = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks": ranks,
             "tile_gather_route": "torch", "c_abi": None,
             "distinct_devices": len({(r["host"], r["device"] if r["pci_bus_id"] is None else r["pci_bus_id"]) for r in ranks}),
             "all_gather_ms": round(float(tm.item()), 4), "all_gather_bytes_per_rank": max_rows * W * 4 * 4, "all_gather_bytes_assembled": world * max_rows * W * 4 * 4,
             "all_gather_timing": "hipEvents around dist.gather_tiles on torch's current stream (the collective is ordered on it), median of 10, max over ranks",
             "frame_checksum_rank0": int(cs0.item()), "frame_equal_across_ranks": bool(flags[0].item()),
             "neighbour_tile_recomputed_equal": bool(flags[1].item())}
    return block, local, full


def worker(args) -> None:
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    solo = os.environ.get("BENCH_SOLO_RANK") == "1"                     # this rank's share of a `world`-rank run, alone on the GPU
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    import torch.distributed as dist
    # one process per GPU; BENCH_BACKEND=gloo + fewer GPUs than ranks is only for rehearsing the multi-rank plumbing
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if world > ndev and backend == "nccl" and not solo:
        raise SystemExit(f"{world} ranks but {ndev} GPU(s): RCCL needs one GPU per rank (BENCH_BACKEND=gloo rehearses the plumbing)")
    dev = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(dev)
    use_dist = (world > 1 and not solo) or os.environ.get("BENCH_FORCE_DIST") == "1"      # BENCH_FORCE_DIST: a 1-rank RCCL group on a one-GPU box
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")       # where the timing all-reduce lives

    from nerf_pytorch_paeng_amd import dist as mdist
    from nerf_pytorch_paeng_amd import nerf_process as NP
    from nerf_pytorch_paeng_amd import ops, synthetic, weights

    def barrier():
        if use_dist:
            dist.barrier()

    def max_over_ranks(t: float) -> float:
        if not use_dist:
            return t
        tm = torch.tensor([t], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        return float(tm.item())

    train_steps = args.train_steps if args.train_steps is not None else (8 if world == 1 else 0)

    # ---- inputs (resident in HBM before any timed region) ----------------------------------------------
    sd = synthetic.make_state_dict(0, 8, 256)
    packed = weights.PackedNeRF.from_state_dict(sd, dev)
    fern = args.workload == "fern"
    K, H, W = synthetic.fern_camera() if fern else synthetic.lego_camera()
    pose = synthetic.fern_pose() if fern else synthetic.pose_spherical(0.0, -30.0, 4.0)
    opts = SimpleNamespace(near=0.0 if fern else 2.0, far=1.0 if fern else 6.0, N_samples_c=SC, N_samples_f=NF, perturb=1.0,
                           chunk_rays=N_RAYS, chunk_pts=524288, data_type="llff" if fern else "blender",
                           gpu_ids=list(range(world)), rank=rank)
    cfg = ops.render_cfg(opts.near, opts.far, SC, NF, False, args.bf16)
    blobs = packed.bf16() if args.bf16 else (packed.coarse, packed.fine)
    pix_all = synthetic.pixel_batch(H, W, N_RAYS * world, 0)

    def make_batch(first: int, n: int):
        """Rays [first, first+n) of the synthetic pixel list, their jitter (keyed on the GLOBAL ray index) and buffers."""
        pix = torch.from_numpy(pix_all[first:first + n]).to(dev)
        o, d = ops.make_o_d_pixels(W, H, K, pose, pix)
        if fern:                                                            # NDC warp, near plane 1 (nerf_process.py:224-226)
            o, d = ops.ndc_rays(H, W, float(K[0][0]), 1.0, o, d)
        b = SimpleNamespace(n=n, first=first, o=o, d=d, rays=torch.cat([o, d], -1).contiguous(),
                            cfg=ops.render_cfg(opts.near, opts.far, SC, NF, False, args.bf16, seed=0, ray_offset=first),
                            cfg16=ops.render_cfg(opts.near, opts.far, SC, NF, False, True, seed=0, ray_offset=first),
                            out=(torch.empty(n, 3, device=dev), torch.empty(n, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, device=dev)),
                            ws=torch.empty(ops.workspace_layout(cfg, n).total, dtype=torch.uint8, device=dev))
        return b

    def step(b):
        # t_rand = u = None: the step draws its own jitter (the reference: torch.rand inside pre_process / sample_pdf) in the sampling
        # kernels, keyed on (seed 0, global ray index b.first + ray, sample): same values every step, so the outputs are reproducible
        ops.render_rays(packed.net, blobs[0], blobs[1], b.cfg, b.rays, None, None, workspace=b.ws, out=b.out)

    def timed_steps(b, warmup: int, steps: int) -> float:
        """W untimed + EXACTLY K timed steps, barrier + synchronize on both sides, max over ranks (seconds).
        In front of the W warm-up steps the same step runs for PREWARM_S of wall time (untimed, like them): the GPU's clock governor needs
        tens of milliseconds to come up from idle, and W = 5 steps of a 512-ray shard (what a rank renders at N = 8) are 5 ms -- measured
        on one GPU: 1.127 ms per step behind 5 warm-up steps, 1.064 behind 50 (profiles/r04_prewarm_512_ray_shard.txt).  A 4096-ray step
        (42 ms of warm-up at W = 5) does not move."""
        t_pw = time.perf_counter()
        while time.perf_counter() - t_pw < PREWARM_S:
            for _ in range(4):
                step(b)
            torch.cuda.synchronize(dev)
        for _ in range(warmup):
            step(b)
        torch.cuda.synchronize(dev)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(b)
        torch.cuda.synchronize(dev)
        barrier()
        return max_over_ranks(time.perf_counter() - t0)

    def median_step_ms(b, steps: int) -> float:
        """SURVEY 8(d)'s protocol beside the contract's mean: `steps` more steps, EACH bracketed by hipEvents recorded on the launch stream
        (torch's current stream is the one the library launches on), median, max over ranks.  Runs behind timed_steps (clock already up);
        the headline `value` does not come from here."""
        import statistics
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for e0, e1 in evs:
            e0.record()
            step(b)
            e1.record()
        torch.cuda.synchronize(dev)
        return max_over_ranks(statistics.median(e0.elapsed_time(e1) for e0, e1 in evs))

    # ---- rays/sec on the 4096-ray batch ------------------------------------------------------------------
    first_s, n_s = mdist.shard_range(N_RAYS, world, rank)            # strong: this rank's contiguous slice of THE batch
    strong = make_batch(first_s, n_s)
    weak = strong if world == 1 else make_batch(rank * N_RAYS, N_RAYS)
    value = value_weak = ms_strong = ms_weak = ms_median = None
    if args.scaling in ("strong", "both") or world == 1:
        el = timed_steps(strong, args.warmup, args.steps)
        value, ms_strong = N_RAYS * args.steps / el, 1e3 * el / args.steps
        ms_median = median_step_ms(strong, max(args.steps, 3))
        assert torch.isfinite(strong.out[2]).all()
    if world == 1:
        value_weak, ms_weak = value, ms_strong
    elif args.scaling in ("weak", "both"):
        el = timed_steps(weak, args.warmup, args.steps)
        value_weak, ms_weak = world * N_RAYS * args.steps / el, 1e3 * el / args.steps
        if ms_median is None:
            ms_median = median_step_ms(weak, max(args.steps, 3))
        assert torch.isfinite(weak.out[2]).all()
    headline_strong = value is not None
    main = strong if headline_strong else weak

    # ---- roofline of the dominant kernel: the fine-network fused MLP launch of the timed step ------------------
    views = ops.workspace_views(cfg, main.n, main.ws)
    z_f = views["z_f"].clone()
    raw_f = torch.empty(main.n, SC + NF, 4, device=dev)
    iters = max(5, min(args.steps, 50))
    ops.time_mlp_rays(packed.net, blobs[1], main.rays, z_f, raw_f, 2, args.bf16)
    k_ms = ops.time_mlp_rays(packed.net, blobs[1], main.rays, z_f, raw_f, iters, args.bf16)
    k_flop = main.n * (SC + NF) * FLOP_PER_POINT                             # algorithmic FLOP per launch
    achieved = k_flop / (k_ms * 1e-3) / 1e12
    # HBM bytes per launch come from separate rocprofv3 --pmc passes (profiles/README.md): a committed figure, tied to the
    # kernel build it was measured on.  It is reported only for the launch shape it was measured at.
    traffic = traffic_build = None
    build_id = kernel_build_id()
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            tj = json.load(f)
        if not args.bf16 and main.n == N_RAYS and not fern:
            traffic, traffic_build = tj["hbm_bytes_per_launch"], tj.get("kernel_build")
    except (OSError, KeyError, ValueError):
        pass
    peak = PEAK_BF16_MFMA_TFLOPS if args.bf16 else PEAK_F32_MFMA_TFLOPS
    roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": traffic,
                "kernel": ("mlp_bf16_kernel" if args.bf16 else "mlp_fp32_kernel<256,0,10,4>") + f" (fine net, {main.n * (SC + NF)} points/launch)",
                "kernel_ms": round(k_ms, 4), "flop_per_launch": k_flop, "kernel_build": build_id,
                "traffic_measured_on_build": traffic_build, "traffic_is_current": (traffic_build == build_id) if traffic is not None else None}
    if args.bf16:
        roofline["peak_is"] = "dense bf16 MFMA"
        roofline["frac_of_f32_mfma_peak"] = round(achieved / PEAK_F32_MFMA_TFLOPS, 4)
        roofline["weight_stream"] = bf16_stream(blobs[1], main.n * (SC + NF), k_ms)

    # ---- BASELINE config #5 on the same shard: the bf16 MFMA variant, timed like the headline and held against the fp32 outputs ----
    bf16_leg = None
    if not args.bf16 and not args.no_bf16_leg:
        cfg16 = ops.render_cfg(opts.near, opts.far, SC, NF, False, True)
        blobs16 = packed.bf16()
        out16 = tuple(torch.empty_like(t) for t in main.out)

        def step16(b=main, out=out16):
            ops.render_rays(packed.net, blobs16[0], blobs16[1], b.cfg16, b.rays, None, None, workspace=b.ws, out=out)

        step(main)                                           # fp32 outputs of this shard as the yardstick (same rays, same jitter)
        ref32 = tuple(t.clone() for t in main.out)
        for _ in range(args.warmup):
            step16()
        torch.cuda.synchronize(dev)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step16()
        torch.cuda.synchronize(dev)
        barrier()
        el16 = max_over_ranks(time.perf_counter() - t0)
        zf16 = ops.workspace_views(cfg16, main.n, main.ws)["z_f"].clone()
        raw16 = torch.empty(main.n, SC + NF, 4, device=dev)
        ops.time_mlp_rays(packed.net, blobs16[1], main.rays, zf16, raw16, 2, True)
        k16 = ops.time_mlp_rays(packed.net, blobs16[1], main.rays, zf16, raw16, iters, True)

        def psnr(x, y):
            return float(-10.0 * torch.log10(torch.mean((x - y) ** 2).clamp_min(1e-20)))

        n_total = N_RAYS if headline_strong else world * N_RAYS

        def quality(out, ref):
            return {"psnr_rgb_c_vs_fp32_dB": round(psnr(out[0], ref[0]), 2), "psnr_rgb_f_vs_fp32_dB": round(psnr(out[2], ref[2]), 2),
                    "max_abs_rgb_f_diff": round(float((out[2] - ref[2]).abs().max()), 5),
                    "rays_beyond_1_grey_level_rgb_f": int(((out[2] - ref[2]).abs().amax(-1) > 1.0 / 255.0).sum()), "rays": int(ref[2].shape[0])}

        # the variant that gives the bf16 frame the fp32 path's DEPTHS: coarse network in split precision (fp32-grade weights_c -> the same
        # fine sample positions as the fp32 path, 64 of a ray's 256 evaluations), fine network in bf16 (MI_NERF_MODE_F16S_BF16)
        cfg_mix = ops.render_cfg(opts.near, opts.far, SC, NF, False, True, seed=0, ray_offset=main.first, coarse_f16s=True)
        blobs_mix = (packed.f16s()[0], blobs16[1])
        out_mix = tuple(torch.empty_like(t) for t in main.out)

        def step_mix(pk=packed, blobs=blobs_mix, out=out_mix):
            ops.render_rays(pk.net, blobs[0], blobs[1], cfg_mix, main.rays, None, None, workspace=main.ws, out=out)

        for _ in range(args.warmup):
            step_mix()
        torch.cuda.synchronize(dev)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_mix()
        torch.cuda.synchronize(dev)
        barrier()
        el_mix = max_over_ranks(time.perf_counter() - t0)

        # config #5's quality figure on a network that is NOT the easy case: the same architecture with the density head x200 (hard
        # surfaces: the coarse pdf is a spike, so a bf16 coarse pass that moves the spike moves every fine sample of the ray) -- the same
        # shard, the same jitter, fp32 / bf16 / mixed.  Trained networks behave like this one, not like Xavier x20
        # (profiles/r05_trained_weights.txt: 47.8-49.5 dB, max |d rgb| 0.145-0.147 on the build's own trained scenes).
        sd_pk = synthetic.make_state_dict(0, 8, 256, density_scale=200.0)
        packed_pk = weights.PackedNeRF.from_state_dict(sd_pk, dev)
        ref_pk = tuple(torch.empty_like(t) for t in main.out)
        b16_pk, mix_pk = tuple(torch.empty_like(t) for t in main.out), tuple(torch.empty_like(t) for t in main.out)
        ops.render_rays(packed_pk.net, packed_pk.coarse, packed_pk.fine, main.cfg,
                        main.rays, None, None, workspace=main.ws, out=ref_pk)
        pk16 = packed_pk.bf16()
        ops.render_rays(packed_pk.net, pk16[0], pk16[1], main.cfg16, main.rays, None, None, workspace=main.ws, out=b16_pk)
        step_mix(packed_pk, (packed_pk.f16s()[0], pk16[1]), mix_pk)
        torch.cuda.synchronize(dev)
        peaked = {"network": "Xavier(seed 0) 8x256 with the density head x200 (bench default: x20): hard surfaces, a spiked coarse pdf",
                  "bf16": quality(b16_pk, ref_pk), "coarse_f16s_fine_bf16": quality(mix_pk, ref_pk),
                  "trained_networks": "profiles/r06_trained_weights_long.txt (20 000-step networks): all-bf16 51.4 / 47.7 dB vs the fp32 frame, max |d rgb| 0.35 / 0.14, "
                                      "and -0.117 dB against ground truth on the 33 dB hard-surface scene (outside the north star's 0.05 dB); coarse_f16s_fine_bf16 "
                                      "67.2 / 62.7 dB, max |d rgb| 0.02, -0.004 dB.  tests/test_gpu_trained.py::test_trained_reduced_precision_frames prints both per scene "
                                      "(profiles/r06_trained_weights.txt: the suite's 3000-step networks)"}
        del packed_pk, pk16, ref_pk, b16_pk, mix_pk

        bf16_leg = {"what": "BASELINE config #5: the same step with bf16 weights / activations on v_mfma_f32_16x16x32_bf16 (fp32 accumulate), this run's shard and jitter. "
                            "NOTE: the coarse network runs in bf16 too, so the FINE SAMPLE POSITIONS differ from the fp32 path's (a moved coarse pdf moves the "
                            "inverse-CDF samples): the PSNR figures here are on Xavier x20 weights, the easy case; `peaked` and `coarse_f16s_fine_bf16` say what "
                            "happens on hard surfaces and what keeping the coarse pass fp32-grade costs",
                    "rays_per_s": round(n_total * args.steps / el16, 1), "ms_per_step": round(1e3 * el16 / args.steps, 4),
                    "fine_kernel_ms": round(k16, 4),
                    "fine_kernel_TFLOPs": round(k_flop / (k16 * 1e-3) / 1e12, 1),
                    "fine_kernel_frac_of_bf16_peak": round(k_flop / (k16 * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
                    "weight_stream": bf16_stream(blobs16[1], main.n * (SC + NF), k16),
                    **{k: v for k, v in quality(out16, ref32).items() if k not in ("rays",)},
                    "coarse_f16s_fine_bf16": {"what": "MI_NERF_MODE_F16S_BF16: coarse network in f16 split precision (fp32-grade: the fine sample positions are the fp32 "
                                                      "path's), fine network in bf16; same shard, same jitter, timed like the leg above",
                                              "rays_per_s": round(n_total * args.steps / el_mix, 1), "ms_per_step": round(1e3 * el_mix / args.steps, 4),
                                              "x_bf16_step_time": round(el_mix / el16, 3),
                                              **{k: v for k, v in quality(out_mix, ref32).items() if k not in ("rays",)}},
                    "peaked": peaked}
        del raw16, zf16

    # ---- the split-precision variant (mlp_f16s.hip): fp32-grade results on the f16 matrix pipe.  An EXTRA leg: `value` and `roofline`
    # above stay the fp32-MFMA kernel's, this one is reported against both peaks with its own dtype name ----
    f16s_leg = None
    if not args.bf16 and not args.no_f16s_leg:
        blobs_s = packed.f16s()
        out_s = tuple(torch.empty_like(t) for t in main.out)
        cfg_s = ops.render_cfg(opts.near, opts.far, SC, NF, False, seed=0, ray_offset=main.first, f16s=True)

        def step_s():
            ops.render_rays(packed.net, blobs_s[0], blobs_s[1], cfg_s, main.rays, None, None, workspace=main.ws, out=out_s)

        step(main)
        ref32 = tuple(t.clone() for t in main.out)
        for _ in range(args.warmup):
            step_s()
        torch.cuda.synchronize(dev)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_s()
        torch.cuda.synchronize(dev)
        barrier()
        el_s = max_over_ranks(time.perf_counter() - t0)
        zf_s = ops.workspace_views(cfg_s, main.n, main.ws)["z_f"].clone()
        raw_s = torch.empty(main.n, SC + NF, 4, device=dev)
        ops.time_mlp_rays(packed.net, blobs_s[1], main.rays, zf_s, raw_s, 2, f16s=True)
        ks = ops.time_mlp_rays(packed.net, blobs_s[1], main.rays, zf_s, raw_s, iters, f16s=True)
        net_tf = k_flop / (ks * 1e-3) / 1e12
        n_total = N_RAYS if headline_strong else world * N_RAYS
        mse = float(torch.mean((out_s[2] - ref32[2]) ** 2))
        f16s_leg = {"what": "the same step with every operand split x = hi + lo 2^-11 into two f16 values and every product taken as hi hi + hi lo + lo hi on "
                            "v_mfma_f32_16x16x32_f16 (fp32 accumulate): fp32-grade results (tests/test_gpu_f16s.py holds it to the fp32 path's parity bars); "
                            "this run's shard and jitter; an extra leg, never the headline",
                    "dtype": "f16 hi+lo split operands, f32 accumulate",
                    "rays_per_s": round(n_total * args.steps / el_s, 1), "ms_per_step": round(1e3 * el_s / args.steps, 4),
                    "fine_kernel_ms": round(ks, 4),
                    "network_TFLOPs": round(net_tf, 1),                                 # the network's arithmetic (what the fp32 kernel's 149 counts)
                    "x_f32_mfma_peak": round(net_tf / PEAK_F32_MFMA_TFLOPS, 3),
                    "issued_f16_mfma_TFLOPs": round(3 * net_tf, 1),                       # three f16 MFMAs per product
                    "issued_frac_of_f16_peak": round(3 * net_tf / PEAK_BF16_MFMA_TFLOPS, 4),  # dense f16 peak = dense bf16 peak (2.5 PF)
                    "speedup_vs_f32_step": round(ms_strong / (1e3 * el_s / args.steps), 3) if headline_strong else None,
                    "max_abs_rgb_c_diff_vs_f32": float((out_s[0] - ref32[0]).abs().max()),
                    "max_abs_rgb_f_diff_vs_f32": float((out_s[2] - ref32[2]).abs().max()),
                    "rays_beyond_1e-4_rgb_f": int(((out_s[2] - ref32[2]).abs().amax(-1) > 1e-4).sum()),
                    "psnr_rgb_f_vs_fp32_dB": round(-10.0 * math.log10(max(mse, 1e-20)), 2)}
        del raw_s, zf_s

    # ---- the same step at small batches (one GPU): what a rank runs under strong scaling ---------------------------
    def small_batch_legs(step_fn, cfg_, blob_fine, is_bf16: bool, peak_tflops: float):
        legs = []
        for n in (256, 512, 1024, 2048):
            b = make_batch(0, n)
            reps = max(args.steps, 50)
            for _ in range(10):                      # a 100 us step: give the clock governor a few ms before timing
                step_fn(b)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(reps):
                step_fn(b)
            torch.cuda.synchronize(dev)
            el = time.perf_counter() - t0
            zf = ops.workspace_views(cfg_, n, b.ws)["z_f"].clone()
            rf = torch.empty(n, SC + NF, 4, device=dev)
            ops.time_mlp_rays(packed.net, blob_fine, b.rays, zf, rf, 2, is_bf16)
            kms = ops.time_mlp_rays(packed.net, blob_fine, b.rays, zf, rf, 20, is_bf16)
            rps = n * reps / el
            legs.append({"rays": n, "ms_per_step": round(1e3 * el / reps, 4), "rays_per_s": round(rps, 1),
                         "frac_of_roofline_end_to_end": round(rps * FLOP_PER_RAY / 1e12 / peak_tflops, 4),
                         "fine_kernel_ms": round(kms, 4),
                         "fine_kernel_frac": round(n * (SC + NF) * FLOP_PER_POINT / (kms * 1e-3) / 1e12 / peak_tflops, 4)})
            del b, zf, rf
        return legs

    small = None
    if world == 1 and not args.no_small_batch:
        small = small_batch_legs(step, cfg, blobs[1], args.bf16, peak)
        if bf16_leg is not None:
            legs16 = small_batch_legs(lambda b: step16(b, b.out), cfg16, blobs16[1], True, PEAK_BF16_MFMA_TFLOPS)
            full16 = bf16_leg["rays_per_s"]
            for leg in legs16:
                leg["frac_of_4096_ray_rate"] = round(leg["rays_per_s"] / full16, 4)
            bf16_leg["small_batch"] = legs16

    # ---- 800x800 frame, rows sharded over the ranks, one all-gather of the tiles ----------------------------
    frame_ms = frame_cs = None
    c_abi_route = use_dist and (backend == "nccl" or bool(os.environ.get("MI_NERF_RCCL_LIB"))) and os.environ.get("BENCH_NO_C_ABI_GATHER") != "1"

    def frame_pose(f: int):
        """Pose of timed frame f (f = -1: the warm-up frame)."""
        if fern:
            return synthetic.fern_pose()
        return pose if f < 0 else synthetic.pose_spherical(3.0 * (f + 1), -30.0, 4.0)

    def frame(fp, via="torch"):
        if solo:                                                        # this rank's row block only; nothing to gather
            return mdist.render_shard(H, W, K, fp, packed, opts, world, rank, seed=0, bf16=args.bf16), None
        return mdist.render_frame(H, W, K, fp, packed, opts, seed=0, bf16=args.bf16, via=via, force_collective=use_dist)

    if args.frames > 0:
        frame(frame_pose(-1))                                           # warm-up frame
        torch.cuda.synchronize(dev)
        barrier()
        t0 = time.perf_counter()
        for f in range(args.frames):
            rgb, disp = frame(frame_pose(f))
        torch.cuda.synchronize(dev)
        barrier()
        frame_ms = 1e3 * max_over_ranks(time.perf_counter() - t0) / args.frames
        assert (rgb.shape == (H, W, 3) or solo) and torch.isfinite(rgb).all()
        if not solo:
            frame_cs = frame_checksum(torch, rgb, disp)                 # of the LAST timed frame: the same pose for any N at the same --frames
        if f16s_leg is not None and not solo:                           # the same frame(s) in split precision, held against the fp32 frame just rendered
            mdist.render_frame(H, W, K, pose, packed, opts, seed=0, f16s=True)
            torch.cuda.synchronize(dev)
            barrier()
            t0 = time.perf_counter()
            for f in range(args.frames):
                rgb_s, _ = mdist.render_frame(H, W, K, frame_pose(f), packed, opts, seed=0, f16s=True)
            torch.cuda.synchronize(dev)
            barrier()
            f16s_leg["frame_ms"] = round(1e3 * max_over_ranks(time.perf_counter() - t0) / args.frames, 2)
            f16s_leg["frame_psnr_vs_f32_dB"] = round(-10.0 * math.log10(max(float(torch.mean((rgb_s - rgb) ** 2)), 1e-20)), 2)
            f16s_leg["frame_pixels_beyond_1_grey_level"] = int(((rgb_s - rgb).abs().amax(-1) > 1.0 / 255.0).sum())

    # ---- N > 1: what the collective saw, so that the first multi-GPU run verifies itself from the driver's record ------------------------
    collective = None
    if use_dist and args.frames > 0 and os.environ.get("BENCH_NO_COLLECTIVE_BLOCK") != "1":
        collective, tile, gathered = collective_block(dist, mdist, torch, dev, backend, rank, world, H, W, K, pose, packed, opts, args.bf16, rgb, disp)
        if c_abi_route:
            # both routes in ONE run: the frames just timed through torch.distributed, now through the library's own RCCL communicator
            collective["c_abi"] = c_abi_route_leg(dist, mdist, torch, dev, coll_dev, tile, gathered, H, W, args.frames,
                                                  lambda fp: frame(fp, via="c_abi"), frame_pose, frame_cs)
        del tile, gathered
    hung = _c_abi_leg_hung            # from here on a hung rank makes no GPU call and no process-group call: it prints (rank 0) and leaves

    # ---- training step: forward + backward + Adam on this rank's 4096-ray batch --------------
    train = None
    if train_steps > 0 and not args.bf16 and not hung:
        from nerf_pytorch_paeng_amd.model import NeRF, get_positional_encoder
        model = NeRF(8, 256, 63, 27).to(dev)
        model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
        posenc = get_positional_encoder(10), get_positional_encoder(4)
        optim = torch.optim.Adam(model.parameters(), lr=5e-4, betas=(0.9, 0.999))          # main.py:79-80
        target = torch.rand(N_RAYS, 3, generator=torch.Generator().manual_seed(rank)).to(dev)
        NP.manual_seed(0)

        def train_step(f16s=False):
            rgb_c, _, rgb_f, _ = NP.batchify_rays_and_render_by_chunk(weak.o, weak.d, model, posenc, H, W, K, opts, ray_offset=rank * N_RAYS, f16s=f16s)
            optim.zero_grad()
            loss = torch.nn.functional.mse_loss(rgb_c, target) + torch.nn.functional.mse_loss(rgb_f, target)   # train.py:60-66
            loss.backward()
            optim.step()
            return loss

        for _ in range(2):
            train_step()
        torch.cuda.synchronize(dev)
        barrier()
        t0 = time.perf_counter()
        for _ in range(train_steps):
            loss = train_step()
        torch.cuda.synchronize(dev)
        barrier()
        tt = max_over_ranks(time.perf_counter() - t0)
        assert torch.isfinite(loss).all()
        t_ms = 1e3 * tt / train_steps
        train_flop_per_ray = 2 * (593408 + 557696 + 593408) * POINTS_PER_RAY            # forward + backward-data + backward-weights
        train = {"ms_per_step": round(t_ms, 3), "rays_per_s": round(world * N_RAYS / (t_ms * 1e-3), 1), "steps": train_steps,
                 "frac_of_f32_mfma_roofline": round(N_RAYS / (t_ms * 1e-3) * train_flop_per_ray / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                 "peak_mem_GiB": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
                 "what": "batchify_rays_and_render_by_chunk (grad) + MSE(rgb_c)+MSE(rgb_f) + loss.backward() + Adam.step(), "
                         f"{N_RAYS} rays per GPU, each rank an independent replica (the reference has no data-parallel training)"}
        # the same step with its three MFMA kernels in split precision (forward with stash, backward-data chain, the wide weight-gradient
        # products: f16 hi + lo operands, fp32 accumulate, fp32-grade gradients): an extra leg, like f16_split for inference -- ms_per_step /
        # frac above stay the all-fp32-MFMA step's
        if not args.no_f16s_leg:
            for _ in range(2):
                train_step(True)
            torch.cuda.synchronize(dev)
            barrier()
            t0 = time.perf_counter()
            for _ in range(train_steps):
                loss = train_step(True)
            torch.cuda.synchronize(dev)
            barrier()
            ts_ms = 1e3 * max_over_ranks(time.perf_counter() - t0) / train_steps
            assert torch.isfinite(loss).all()
            train["f16_split"] = {"ms_per_step": round(ts_ms, 3), "rays_per_s": round(world * N_RAYS / (ts_ms * 1e-3), 1),
                                  "speedup_vs_f32_step": round(t_ms / ts_ms, 3),
                                  "x_f32_mfma_roofline": round(N_RAYS / (ts_ms * 1e-3) * train_flop_per_ray / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                  "dtype": "f16 hi+lo split operands, f32 accumulate",
                                  "what": "batchify_rays_and_render_by_chunk(..., f16s=True): mlp_f16s_kernel<STASH>, dgrad_f16s_kernel, wgrad_f16s_kernel in place of "
                                          "the three fp32-MFMA kernels (same stash, deltas and gradient layout; narrow products, compositing and Adam unchanged); "
                                          "gradients within the fp32 path's bars of an fp64 evaluation (tests/test_gpu_train.py F11, f16s case)"}
        # roofline leg of the training kernels' GEMM: the 256x256 weight-gradient products over the fine net's 786 432 points, as the
        # backward pass runs them (nine in one launch: mi_nerf_wgrad_products), and the same entry with one product
        n_pts = N_RAYS * (SC + NF)
        dlt = torch.randn(n_pts + 64, 256, device=dev)
        xin = torch.randn(n_pts + 64, 256, device=dev)
        ops.wgrad_product(dlt, 256, xin, 256, n_pts, iters=2)
        _, _, wg_ms = ops.wgrad_product(dlt, 256, xin, 256, n_pts, iters=10, timed=True)
        wg_tf = 2.0 * 256 * 256 * n_pts / (wg_ms * 1e-3) / 1e12
        ops.wgrad_products([dlt] * 9, [xin] * 9, n_pts, iters=1)
        _, _, wg9_ms = ops.wgrad_products([dlt] * 9, [xin] * 9, n_pts, iters=5, timed=True)
        wg9_tf = 9 * 2.0 * 256 * 256 * n_pts / (wg9_ms * 1e-3) / 1e12
        train["wgrad_256x256"] = {"ms": round(wg_ms, 4), "achieved_TFLOPs": round(wg_tf, 1), "frac_of_f32_mfma_peak": round(wg_tf / PEAK_F32_MFMA_TFLOPS, 4),
                                  "what": "ONE product in a launch of its own (mi_nerf_wgrad_products, n = 1: wgrad_big_kernel over 256 point slices + reduce; hipEvents on the launch stream) -- nothing on the path runs one alone, and the C ABI has no single-product entry since round 4"}
        if not args.no_f16s_leg:                                         # the same nine products in split precision: bound by their HBM reads
            ops.wgrad_products([dlt] * 9, [xin] * 9, n_pts, iters=1, f16s=True)
            _, _, wgs_ms = ops.wgrad_products([dlt] * 9, [xin] * 9, n_pts, iters=5, timed=True, f16s=True)
            read_gb = 9 * (2 + 1) * n_pts * 256 * 4 / 1e9               # per product: both operands once + the scale-finding pass over the gradient operand
            train["wgrad_9x256x256_f16_split"] = {
                "ms": round(wgs_ms, 4), "ms_per_product": round(wgs_ms / 9, 4), "network_TFLOPs": round(9 * 2.0 * 256 * 256 * n_pts / (wgs_ms * 1e-3) / 1e12, 1),
                "x_f32_mfma_peak": round(9 * 2.0 * 256 * 256 * n_pts / (wgs_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 3),
                "roofline": {"bound": "hbm", "achieved": round(read_gb / (wgs_ms * 1e-3), 1), "peak": 8000.0, "unit": "GB/s", "frac": round(read_gb / (wgs_ms * 1e-3) / 8000.0, 4),
                             "algorithmic_GB": round(read_gb, 2)},
                "what": "mi_nerf_wgrad_products_f16s: nine products in one launch with operands converted on the fly to f16 hi + lo pairs, dense random operands, "
                        "including the pass over each gradient operand that finds its scale (the training step takes the scale from d_raw instead and its "
                        "operands are ReLU-sparse: 2.9 ms per fine-net batch there)"}
        train["wgrad_9x256x256"] = {"ms": round(wg9_ms, 4), "ms_per_product": round(wg9_ms / 9, 4), "achieved_TFLOPs": round(wg9_tf, 1),
                                    "frac_of_f32_mfma_peak": round(wg9_tf / PEAK_F32_MFMA_TFLOPS, 4),
                                    "what": "nine products in one launch (mi_nerf_wgrad_products), the form the backward pass uses for a net's wide layers"}
        del model, optim, dlt, xin
        torch.cuda.empty_cache()

    # ---- global-batch staging: rays for 100 800x800 training images + epoch shuffle, on the device ----
    staging = None
    if rank == 0 and world == 1 and not args.bf16 and not fern and train_steps > 0 and not hung:
        from nerf_pytorch_paeng_amd import harness
        n_img = 100
        imgs = torch.rand(n_img, H, W, 3, device=dev)
        poses_tr = torch.from_numpy(np.stack([synthetic.pose_spherical(3.6 * i - 180.0, -30.0, 4.0) for i in range(n_img)], 0)).float().to(dev)
        harness.global_batch(imgs[:2], K, poses_tr[:2], [0, 1], (H, W), dev)                    # warm-up
        torch.cuda.synchronize(dev)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        p34 = poses_tr[:, :3, :4].contiguous()
        perm = torch.randperm(n_img * H * W, device=dev)          # torch's generator: plumbing, not timed
        e0.record()
        rr = ops.rays_rgb(W, H, K, p34, imgs)
        e1.record()
        rr2 = ops.permute_rows(rr, perm)                          # the reference's form (np.random.shuffle of the table): for the record only
        e2.record()
        torch.cuda.synchronize(dev)
        # what the training loop does since round 6: the table stays unshuffled, a step gathers its B rows through the permutation
        import statistics
        shuffled = harness.ShuffledRows(rr, perm)
        gev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
        for i, (g0, g1) in enumerate(gev):
            g0.record()
            batch = shuffled[i * N_RAYS:(i + 1) * N_RAYS]
            g1.record()
        torch.cuda.synchronize(dev)
        assert torch.equal(batch, rr2[49 * N_RAYS:50 * N_RAYS])
        gather_us = 1e3 * statistics.median(g0.elapsed_time(g1) for g0, g1 in gev[5:])
        n_rays_gb = n_img * H * W
        b_gen = n_rays_gb * (12 + 36)                     # read the pixel, write (o, d, rgb)
        b_perm = n_rays_gb * (36 + 36 + 8)                # gather + write + the permutation itself
        # a random 36-byte row is one 128-byte HBM request, two when it straddles a line (offsets 96..124 of 128: 8 of 32 four-byte positions): MEASURED,
        # TCC_EA0_RDREQ = 1.25 per row exactly (profiles/r06_staging_pmc_req.json, r06_permute_rows_bound.txt); the permutation streams (8 B per row)
        b_perm_hbm = n_rays_gb * (128 * 1.25 + 36 + 8)
        staging = {"what": f"main.py:92-102 on the device: rays_rgb for {n_img} {H}x{W} images ({n_rays_gb * 36 / 1e9:.2f} GB); the epoch shuffle is a permutation "
                           f"held beside the table (harness.ShuffledRows), a step gathers its {N_RAYS} rows through it (train.py:29)",
                   "rays_rgb_ms": round(e0.elapsed_time(e1), 3), "rays_rgb_GBps": round(b_gen / e0.elapsed_time(e1) / 1e6, 1),
                   "batch_gather_us": round(gather_us, 2), "batch_gather_rows": N_RAYS,
                   "batch_gather_is": "ops.gather_rows (mi_nerf_permute_rows, n = B) of one step's rows from the unshuffled table, hipEvents on the launch stream, "
                                      "median of 45; launch-bound (147 KB moved); per epoch: 15 625 steps x this instead of one shuffle_ms and a second 2.3 GB table",
                   "materialized_shuffle": {"ms": round(e1.elapsed_time(e2), 3), "GBps_algorithmic": round(b_perm / e1.elapsed_time(e2) / 1e6, 1),
                                            "GBps_hbm_traffic": round(b_perm_hbm / e1.elapsed_time(e2) / 1e6, 1),
                                            "hbm_traffic_is": "this run's time x the bytes the PMC passes counted for this kernel and table (1.25 128-byte read requests per "
                                                              "36-byte row, every written byte once: profiles/r06_staging_pmc_req.json, r06_staging_pmc_write.json)",
                                            "why_not_used": "a random 36-byte row costs a 128-byte HBM request (1.25 measured): 80 algorithmic bytes per row move 204, the "
                                                            "kernel runs at the HBM rate in the bytes it really moves, and 8 TB/s caps the algorithmic rate at 3.1 TB/s "
                                                            "whatever the kernel; the training loop no longer materialises the shuffle (profiles/r06_permute_rows_bound.txt)"},
                   "hbm_peak_GBps": 8000}
        del imgs, rr, rr2, shuffled, batch
        torch.cuda.empty_cache()

    # ---- CPU baseline: the oracle (a port: the reference cannot leave the build container) -------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not hung:
        from oracle import restate as R
        # the GPU box gives one GPU's job a 16-core share of a many-core host: do not oversubscribe
        n_thr = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
        torch.set_num_threads(n_thr)
        # the same jitter the GPU step draws in its kernels: (seed 0, global ray index, sample)
        rc = main.rays.cpu()
        tc, uc = ops.fill_uniform(0, 0, main.first, main.n, SC, dev).cpu(), ops.fill_uniform(0, 1, main.first, main.n, NF, dev).cpu()
        pcfg = R.PathConfig()
        with torch.no_grad():
            R.render_rays(rc[:256], sd, pcfg, tc[:256], uc[:256])                            # warm-up
            # the WHOLE 4096-ray batch per call, the reference's own chunk size (chunk_rays = 4096): a 1024-ray sample
            # (rounds 1-3) gave the CPU smaller GEMMs per call than the reference runs and read 30 % low
            reps, spent, n_cpu = [], 0.0, main.n
            while len(reps) < 3:                             # >= 3 reps, median (3 x ~6.5 s on the GPU box's 16-core share)
                t0 = time.perf_counter()
                R.render_rays(rc[:n_cpu], sd, pcfg, tc[:n_cpu], uc[:n_cpu])
                dt = time.perf_counter() - t0
                reps.append(dt)
                spent += dt
        train_cpu = None
        if train is not None:                                   # same port, one training step (autograd) on a bounded sample
            n_tr = 256
            psd = {k: torch.as_tensor(v).clone().float().requires_grad_(True) for k, v in sd.items()}
            tgt = torch.rand(n_tr, 3, generator=torch.Generator().manual_seed(0))
            t0 = time.perf_counter()
            ref = R.render_rays(rc[:n_tr], psd, pcfg, tc[:n_tr], uc[:n_tr])
            (torch.mean((ref["rgb_c"] - tgt) ** 2) + torch.mean((ref["rgb_f"] - tgt) ** 2)).backward()
            train_cpu = round(n_tr / (time.perf_counter() - t0), 1)
        cpu = {"value": round(n_cpu / float(np.median(reps)), 1), "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
               "train_rays_per_s": train_cpu,
               "sample": f"oracle/restate.py render_rays (torch CPU fp32, {torch.get_num_threads()} threads: this job's share of the host) on the first "
                         f"{n_cpu} rays of the same 4096-ray batch (the reference's chunk size), median of {len(reps)} rep(s), {spent:.0f} s of CPU work; "
                         "SURVEY section 6 timed the reference itself at 691 rays/s on 8 vCPU of the build container"}

    if rank == 0 or solo:
        head_value = value if headline_strong else value_weak
        head_ms = ms_strong if headline_strong else ms_weak
        per_gpu = n_s if headline_strong else N_RAYS
        line = {
            "metric": baseline_metric(),
            "value": round(head_value, 1), "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(head_ms, 4), "ms_per_step_median": None if ms_median is None else round(ms_median, 4),
            "ms_per_step_median_is": "median over max(K, 3) further steps, each bracketed by hipEvents on the launch stream, max over ranks (SURVEY 8(d)); "
                                     "`value` is the contract's K steps between two barriers + synchronisations",
            "higher_is_better": True, "scaling": "strong" if headline_strong else "weak", "vs_baseline": None,
            "dtype": "bf16" if args.bf16 else "f32", "data": "synthetic",
            "value_weak": None if value_weak is None else round(value_weak, 1),
            "ms_per_step_weak": None if ms_weak is None else round(ms_weak, 4),
            "config": {"workload": ("fern LLFF NDC rays, coarse+fine 4096-ray batch, 64+128 samples, 8x256 MLP (BASELINE config #4)" if fern else
                                    "lego coarse+fine 4096-ray batch, 64+128 samples, 8x256 MLP (BASELINE config #2)") +
                                   (" -- bf16 MFMA variant (config #5)" if args.bf16 else ""),
                       "rays_per_gpu": per_gpu, "rays_per_gpu_weak": N_RAYS, "samples": [SC, NF], "net": "8x256, skip 4, L_x 10, L_d 4",
                       "prewarm_ms": round(1e3 * PREWARM_S, 1),
                       "parallelism": (f"the 4096-ray batch split into {world} contiguous slices of {per_gpu} rays, one per GPU, no data-path collective "
                                       f"(value); value_weak: {N_RAYS} rays on each of {world} GPU(s); frame: rows over {world} GPU(s) + one all-gather")},
            "frame_ms_800x800": None if (frame_ms is None or fern) else round(frame_ms, 2),
            "frame_ms": None if frame_ms is None else round(frame_ms, 2), "frame_hw": [H, W],
            "frame_checksum": frame_cs,                # bit patterns of the last timed frame: compare across N (and with collective.frame_checksum_rank0)
            "frac_of_roofline_end_to_end": round(head_value / world * FLOP_PER_RAY / 1e12 / peak, 4),
            "roofline": roofline,
        }
        if collective is not None:
            line["collective"] = collective
            ca = collective.get("c_abi") or {}
            # the same frames through the library's own RCCL communicator (mi_nerf_all_gather_tiles), timed in this run beside frame_ms
            line["frame_ms_c_abi"] = ca.get("frame_ms")
            line["frame_checksum_c_abi"] = ca.get("frame_checksum")
        if solo:
            line["solo_rank"] = {"rank": rank, "of": world, "n_gpus_measured": 1, "what": "BENCH_SOLO_RANK=1: this rank's share of the run timed alone (no process group, "
                                 "no gather); `value` assumes every rank takes as long as this one"}
        if not args.bf16:
            line["frac_of_f32_mfma_roofline_end_to_end"] = line["frac_of_roofline_end_to_end"]
        if bf16_leg is not None:
            line["bf16"] = bf16_leg
        if f16s_leg is not None:
            line["f16_split"] = f16s_leg
        if small is not None:
            line["small_batch"] = small
        if train is not None:
            line["train"] = train
        if staging is not None:
            line["staging"] = staging
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line, ensure_ascii=False), flush=True)
    status = exit_status(hung, backend, world, collective)
    if status == EXIT_C_ABI_HUNG:      # its side stream cannot be drained and its communicator cannot be destroyed: no teardown, a status CI can see
        sys.stdout.flush()
        sys.stderr.write(f"bench.py: rank {rank}: the C-ABI gather leg hung ({collective['c_abi']['error']})\n")
        sys.stderr.flush()
        os._exit(status)
    if use_dist:
        barrier()                      # rank 0 may still be printing / staging: tear the group down together
        mdist.close_tile_comms()
        dist.destroy_process_group()
    if status:                         # RCCL must have seen N GPUs: N ranks on fewer devices measure nothing about N GPUs
        sys.stderr.write(f"bench.py: backend nccl with {world} rank(s) but {collective['distinct_devices']} distinct device(s) in collective.ranks\n")
        sys.exit(status)


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_workers(args.gpus))             # before anything in this process touches the GPU
    worker(args)


if __name__ == "__main__":
    main()
