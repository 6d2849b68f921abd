"""x-slab sharding of the pixel grid across ranks and the two exchange steps of
the path:

  C2  all-reduce(sum) of per-rank partial sums for the pixel means
      (math_tools.rs:421-440) — a few KB
  C1  gather of the per-rank image slabs to rank 0 (what the GUI reads after a
      recompute, data_thread.rs:1310-1315)

Every trace is independent, so the compute itself needs no collective.  The
split mirrors the reference's rayon split over Axis(0): contiguous x rows.
Backend-agnostic: `nccl` (= RCCL over xGMI) on GPUs, `gloo` in the CPU tests.
"""
from __future__ import annotations

from typing import Optional, Tuple


def slab(nx: int, world: int, rank: int) -> Tuple[int, int]:
    """(x0, nx_local) of `rank`: contiguous rows, remainder spread over the first ranks."""
    base, rem = divmod(nx, world)
    n = base + (1 if rank < rem else 0)
    x0 = rank * base + min(rank, rem)
    return x0, n


def all_reduce_sums(t_sums, dist=None):
    """C2: in-place sum of the partial-sum vector over ranks."""
    if dist is not None and dist.is_initialized():
        dist.all_reduce(t_sums)
    return t_sums


def gather_image(t_img, nx: int, dist=None, dst: int = 0):
    """C1: gathers (nx_local, ny) image slabs into the (nx, ny) image on `dst`.
    Slabs may differ by one row (nx % world != 0), so this is a padded gather."""
    import torch

    if dist is None or not dist.is_initialized():
        return t_img
    world, rank = dist.get_world_size(), dist.get_rank()
    ny = t_img.shape[1]
    rows = [slab(nx, world, r)[1] for r in range(world)]
    mx = max(rows)
    send = t_img
    if t_img.shape[0] != mx:
        send = torch.zeros((mx, ny), dtype=t_img.dtype, device=t_img.device)
        send[: t_img.shape[0]] = t_img
    bufs = [torch.empty((mx, ny), dtype=t_img.dtype, device=t_img.device) for _ in range(world)] \
        if rank == dst else None
    dist.gather(send.contiguous(), bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:n] for b, n in zip(bufs, rows)], dim=0)


def band_range(n_filters: int, world: int, rank: int) -> Tuple[int, int]:
    """Deconvolution is band-parallel across GPUs (Richardson-Lucy is spatially
    global per band): bands [b0, b1) of this rank; every rank
    needs the whole cube; the per-rank outputs are all-reduced (sum)."""
    b0, n = slab(n_filters, world, rank)
    return b0, b0 + n


def all_reduce_cube(t_out, dist=None):
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t_out)
    return t_out
