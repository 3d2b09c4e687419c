"""Multi-GPU frame rendering: shard rays, render locally, assemble image tiles.

The path shards embarrassingly (every ray is independent end to end): one process per
GPU, each renders a contiguous block of image rows generated on its own device from (K, pose, row
range) -- no input scatter, no collective during compute -- followed by ONE all-gather of the
``[rows_local * W, 4]`` fp32 output tile (rgb + disp; 1.28 MB per GPU for 800x800 over 8 GPUs) over
RCCL/xGMI.  The jitter is keyed on the global ray index, so the assembled frame is bit-identical for any
world size.  The reference is single-GPU only (main.py:166-170); this module has no counterpart there.

Two routes to the same collective (``gather_tiles(..., via=)``):

* ``"c_abi"`` -- ``mi_nerf_all_gather_tiles`` of libmi_nerf.so (csrc/comm.hip): an RCCL communicator of the library's own
  (``TileComm``; bootstrapped from a 128-byte unique id that rank 0 creates and any channel distributes -- here a
  ``torch.distributed`` broadcast), the all-gather enqueued on the HIP stream the render ran on, ragged splits padded and
  un-padded by the library.  What a caller of the C ABI that is not PyTorch uses.
* ``"torch"`` -- ``torch.distributed.all_gather_into_tensor`` (backend "nccl" is RCCL on ROCm; "gloo" for CPU rehearsals).
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from ._lib import MiNerfError, check, dev_ptr, lib, stream_ptr

COMM_ID_BYTES = 128          # MI_NERF_COMM_ID_BYTES


def shard_rows(H: int, world: int, rank: int) -> Tuple[int, int]:
    """(row0, n_rows) of this rank's contiguous row block; the first H % world ranks get one extra row."""
    base, extra = divmod(H, world)
    n = base + (1 if rank < extra else 0)
    row0 = rank * base + min(rank, extra)
    return row0, n


def shard_range(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Same split for a flat list of n rays (the 4096-ray batch metric)."""
    return shard_rows(n, world, rank)


class TileComm:
    """An RCCL communicator of libmi_nerf.so for the tile gather (mi_nerf_comm_* / mi_nerf_all_gather_tiles, include/mi_nerf.h).
    ``TileComm(id_bytes, world, rank, device)`` is collective over all ranks; ``TileComm.unique_id()`` makes the id on rank 0.
    ``close()`` destroys the communicator (collective too: call it on every rank, before the process group goes away).
    One gather at a time per TileComm: the staging buffer of a ragged split belongs to the communicator, and RCCL orders a communicator's
    collectives by issue, so gathers of one TileComm go to one stream (the stream the frame is rendered on)."""

    def __init__(self, id_bytes: bytes, world: int, rank: int, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise MiNerfError(f"TileComm lives on a HIP device, got {self.device}")
        if len(id_bytes) != COMM_ID_BYTES:
            raise MiNerfError(f"the unique id has {COMM_ID_BYTES} bytes, got {len(id_bytes)}")
        self.world, self.rank = int(world), int(rank)
        self._staging: Optional[torch.Tensor] = None
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            check(lib().mi_nerf_comm_init_rank(C.c_char_p(bytes(id_bytes)), self.world, self.rank, C.byref(handle)), "mi_nerf_comm_init_rank")
        self._handle = handle

    @staticmethod
    def available() -> None:
        """Raise unless librccl can be resolved (it is loaded on first use); makes no communicator and no bootstrap socket."""
        check(lib().mi_nerf_rccl_available(), "mi_nerf_rccl_available")

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(COMM_ID_BYTES)
        check(lib().mi_nerf_comm_unique_id(buf), "mi_nerf_comm_unique_id")
        return buf.raw

    @classmethod
    def from_group(cls, device, group=None) -> "TileComm":
        """Bootstrap over an initialised ``torch.distributed`` group: rank 0's id travels in one 128-byte broadcast.
        Collective, and safe against a rank that cannot take part: every rank first reports "librccl resolved (and, on rank 0, the id
        made)" into an all-reduce(MIN); when any rank reports 0, EVERY rank raises MiNerfError before the broadcast and before
        ncclCommInitRank, so no rank is left waiting for one that has already given up."""
        device = torch.device(device)
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        carrier = device if dist.get_backend(group) == "nccl" else torch.device("cpu")
        ident, why = bytes(COMM_ID_BYTES), None
        try:
            cls.available()
            if rank == 0:
                ident = cls.unique_id()
        except MiNerfError as e:
            why = e
        ok = torch.tensor([0 if why is not None else 1], dtype=torch.int32, device=carrier)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 0:
            raise MiNerfError(f"TileComm.from_group: RCCL is not usable on every rank of the group (rank {rank}: {why if why is not None else 'ok here'})")
        t = torch.frombuffer(bytearray(ident), dtype=torch.uint8).to(carrier)
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(bytes(t.cpu().numpy().tobytes()), world, rank, device)

    def all_gather_tiles(self, local: torch.Tensor, H: int, W: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``[rows_local * W, C]`` -> ``[H * W, C]`` on every rank, enqueued on the current stream of the tile's device."""
        if self._handle is None:
            raise MiNerfError("the communicator is closed")
        if local.dim() != 2 or local.shape[0] % W:
            raise MiNerfError(f"tile must be [rows_local * {W}, C], got {tuple(local.shape)}")
        Cc = int(local.shape[1])
        rows = local.shape[0] // W
        frame = torch.empty(H * W, Cc, dtype=torch.float32, device=local.device) if out is None else out
        if tuple(frame.shape) != (H * W, Cc) or frame.device != local.device:
            raise MiNerfError(f"out must be [{H * W}, {Cc}] on {local.device}, got {tuple(frame.shape)} on {frame.device}")
        need = int(lib().mi_nerf_all_gather_staging_bytes(self.world, H, W, Cc))
        if need and (self._staging is None or self._staging.numel() < need):
            self._staging = torch.empty(need, dtype=torch.uint8, device=local.device)
        with torch.cuda.device(local.device):
            check(lib().mi_nerf_all_gather_tiles(self._handle, dev_ptr(local, "tile"), rows, H, W, Cc, dev_ptr(frame, "frame"),
                                                 dev_ptr(self._staging, "staging", torch.uint8, 16) if need else None, need,
                                                 stream_ptr(local.device)), "mi_nerf_all_gather_tiles")
        return frame

    def close(self) -> None:
        if self._handle is not None:
            h, self._handle = self._handle, None
            with torch.cuda.device(self.device):
                check(lib().mi_nerf_comm_destroy(h), "mi_nerf_comm_destroy")


def unpad_tiles(staging: torch.Tensor, world: int, H: int, W: int, C_: int) -> torch.Tensor:
    """``staging`` [world, max_rows * W * C] (every rank's tile padded to the largest block, as the ragged all-gather leaves it) ->
    frame [H * W, C]: the copy kernel behind mi_nerf_all_gather_tiles on its own (any world size on one GPU)."""
    frame = torch.empty(H * W, C_, dtype=torch.float32, device=staging.device)
    with torch.cuda.device(staging.device):
        check(lib().mi_nerf_unpad_tiles(dev_ptr(staging, "staging"), int(world), int(H), int(W), int(C_), dev_ptr(frame), stream_ptr(staging.device)),
              "mi_nerf_unpad_tiles")
    return frame


_tile_comms: Dict[Tuple[int, int], TileComm] = {}


def tile_comm(device, group=None) -> TileComm:
    """The process's TileComm for (group, device), created on first use (collective over the group)."""
    device = torch.device(device)
    key = (id(group) if group is not None else 0, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _tile_comms:
        _tile_comms[key] = TileComm.from_group(device, group)
    return _tile_comms[key]


def close_tile_comms() -> None:
    """Destroy every cached TileComm (call on every rank before dist.destroy_process_group())."""
    while _tile_comms:
        _tile_comms.popitem()[1].close()


def gather_tiles(local: torch.Tensor, H: int, W: int, group=None, force_collective: bool = False, via: str = "torch") -> torch.Tensor:
    """All-gather per-rank ``[rows_local * W, C]`` tiles into the full ``[H * W, C]`` frame on every rank.
    A group of one rank returns its tile as is; ``force_collective`` sends it through the collective anyway (a one-GPU box can
    then exercise the RCCL call path itself).  ``via``: "torch" (torch.distributed) or "c_abi" (mi_nerf_all_gather_tiles on the
    current stream; HIP tensors only)."""
    if via not in ("torch", "c_abi"):
        raise ValueError(f"via must be 'torch' or 'c_abi', got {via!r}")
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force_collective):
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if via == "c_abi":
        if not local.is_cuda:
            raise MiNerfError(f"via='c_abi' gathers through the library's RCCL communicator and needs tiles on a HIP device (got {local.device}); there is no host fallback")
        _, n_rows = shard_rows(H, world, rank)
        if local.shape[0] != n_rows * W:
            raise ValueError(f"rank {rank}: tile has {local.shape[0]} rays, expected {n_rows * W}")
        return tile_comm(local.device, group).all_gather_tiles(local.contiguous(), H, W)
    C = local.shape[1]
    max_rows = (H + world - 1) // world
    _, n_rows = shard_rows(H, world, rank)
    if local.shape[0] != n_rows * W:
        raise ValueError(f"rank {rank}: tile has {local.shape[0]} rays, expected {n_rows * W}")
    padded = local
    if n_rows < max_rows:                                   # ragged split: pad to the common tile size
        padded = torch.zeros(max_rows * W, C, dtype=local.dtype, device=local.device)
        padded[:n_rows * W] = local
    dev = local.device
    if dev.type == "cuda" and dist.get_backend(group) == "gloo":     # CPU rehearsal of the multi-rank path
        padded = padded.cpu()
    out = torch.empty(world * max_rows * W, C, dtype=local.dtype, device=padded.device)
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    out = out.to(dev)
    if H % world == 0:
        return out
    parts = []
    for r in range(world):
        _, nr = shard_rows(H, world, r)
        parts.append(out[r * max_rows * W:r * max_rows * W + nr * W])
    return torch.cat(parts, 0)


def assemble_tiles(tiles: Sequence[torch.Tensor], H: int, W: int) -> torch.Tensor:
    """``[H * W, C]`` frame from the per-rank tiles in rank order (what ``gather_tiles`` returns on every rank); checks that
    tile r has exactly the rays of ``shard_rows(H, len(tiles), r)``."""
    world = len(tiles)
    for r, t in enumerate(tiles):
        _, nr = shard_rows(H, world, r)
        if t.shape[0] != nr * W:
            raise ValueError(f"tile {r} of {world} has {t.shape[0]} rays, expected {nr * W}")
    return tiles[0] if world == 1 else torch.cat(list(tiles), 0)


def render_shard(H: int, W: int, K, pose, model, opts, world: int, rank: int, *, seed: int = 0, bf16: bool = False,
                 f16s: bool = False, coarse_f16s: bool = False) -> torch.Tensor:
    """Rank ``rank``'s ``[rows_local * W, 4]`` tile (rgb + disp) of an H x W frame split over ``world`` ranks: rays generated on
    this device from (K, pose, row range), jitter keyed on the GLOBAL ray index (``ray_offset`` = first pixel of the block), the
    fine outputs when ``N_samples_f > 0`` else the coarse ones (test.py:42-47).  No communication."""
    from . import nerf_process as NP
    from . import ops
    from .weights import packed_for
    packed = packed_for(model)
    r0, nr = shard_rows(H, world, rank)
    _, d = ops.make_o_d(W, H, K, pose, packed.device, row0=r0, n_rows=nr, want_origins=False)
    p = pose if isinstance(pose, torch.Tensor) else torch.as_tensor(pose)
    o = p[:3, -1].to(packed.device, torch.float32).expand(d.shape)
    rc, dc, rf, df = NP.batchify_rays_and_render_by_chunk(o, d, packed, None, H, W, K, opts, seed=seed, ray_offset=r0 * W, bf16=bf16, f16s=f16s,
                                                          coarse_f16s=coarse_f16s)
    rgb, disp = (rc, dc) if int(opts.N_samples_f) == 0 else (rf, df)
    return torch.cat([rgb, disp[:, None]], -1)


def render_frame(H: int, W: int, K, pose, model, opts, *, seed: int = 0, group=None, bf16: bool = False, f16s: bool = False,
                 render_rows_fn: Optional[Callable[[int, int], torch.Tensor]] = None, via: str = "torch",
                 force_collective: bool = False, coarse_f16s: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """Render one H x W frame sharded over the process group; returns (rgb [H,W,3], disp [H,W]) on every rank.

    Counterpart of the per-pose body of the reference's test()/render() harness (test.py:38-53,143-152):
    make_o_d -> batchify_rays_and_render_by_chunk -> pick the fine outputs when N_samples_f > 0.
    ``render_rows_fn(row0, n_rows) -> [n_rows*W, 4]`` overrides the local renderer (used by CPU tests); ``via`` / ``force_collective`` as in
    ``gather_tiles``.
    """
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    if render_rows_fn is None:
        local = render_shard(H, W, K, pose, model, opts, world, rank, seed=seed, bf16=bf16, f16s=f16s, coarse_f16s=coarse_f16s)
    else:
        local = render_rows_fn(*shard_rows(H, world, rank))
    full = gather_tiles(local, H, W, group, force_collective=force_collective, via=via)
    return full[:, :3].reshape(H, W, 3), full[:, 3].reshape(H, W)
