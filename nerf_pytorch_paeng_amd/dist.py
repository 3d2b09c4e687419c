"""Multi-GPU frame rendering: shard rays, render locally, assemble image tiles.

The path shards embarrassingly (every ray is independent end to end): one process per
GPU, each renders a contiguous block of image rows generated on its own device from (K, pose, row
range) -- no input scatter, no collective during compute -- followed by ONE all-gather of the
``[rows_local * W, 4]`` fp32 output tile (rgb + disp; 1.28 MB per GPU for 800x800 over 8 GPUs) over
RCCL/xGMI (``torch.distributed`` backend "nccl" on ROCm).  The jitter is keyed on the global ray index,
so the assembled frame is bit-identical for any world size.  The reference is single-GPU only
(main.py:166-170); this module has no counterpart there.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_rows(H: int, world: int, rank: int) -> Tuple[int, int]:
    """(row0, n_rows) of this rank's contiguous row block; the first H % world ranks get one extra row."""
    base, extra = divmod(H, world)
    n = base + (1 if rank < extra else 0)
    row0 = rank * base + min(rank, extra)
    return row0, n


def shard_range(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Same split for a flat list of n rays (the 4096-ray batch metric)."""
    return shard_rows(n, world, rank)


def gather_tiles(local: torch.Tensor, H: int, W: int, group=None, force_collective: bool = False) -> torch.Tensor:
    """All-gather per-rank ``[rows_local * W, C]`` tiles into the full ``[H * W, C]`` frame on every rank.
    A group of one rank returns its tile as is; ``force_collective`` sends it through the collective anyway (a one-GPU box can
    then exercise the RCCL call path itself)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force_collective):
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
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
                 f16s: bool = False) -> torch.Tensor:
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
    rc, dc, rf, df = NP.batchify_rays_and_render_by_chunk(o, d, packed, None, H, W, K, opts, seed=seed, ray_offset=r0 * W, bf16=bf16, f16s=f16s)
    rgb, disp = (rc, dc) if int(opts.N_samples_f) == 0 else (rf, df)
    return torch.cat([rgb, disp[:, None]], -1)


def render_frame(H: int, W: int, K, pose, model, opts, *, seed: int = 0, group=None, bf16: bool = False, f16s: bool = False,
                 render_rows_fn: Optional[Callable[[int, int], torch.Tensor]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Render one H x W frame sharded over the process group; returns (rgb [H,W,3], disp [H,W]) on every rank.

    Counterpart of the per-pose body of the reference's test()/render() harness (test.py:38-53,143-152):
    make_o_d -> batchify_rays_and_render_by_chunk -> pick the fine outputs when N_samples_f > 0.
    ``render_rows_fn(row0, n_rows) -> [n_rows*W, 4]`` overrides the local renderer (used by CPU tests).
    """
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    if render_rows_fn is None:
        local = render_shard(H, W, K, pose, model, opts, world, rank, seed=seed, bf16=bf16, f16s=f16s)
    else:
        local = render_rows_fn(*shard_rows(H, world, rank))
    full = gather_tiles(local, H, W, group)
    return full[:, :3].reshape(H, W, 3), full[:, 3].reshape(H, W)
