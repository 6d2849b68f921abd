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


# ---- 3-D voxel envelope across x-slabs (gui/threed_plot.rs:205-214) ------------------------------
SELECT_FLOOR_BIN = (0x80000000 | 0x3A800000) >> 21   # keys below 2^-10 are lumped at level 0 (voxel_api.cpp)


def select_kth_largest(local_hist, k: int, dist=None, device=None):
    """Radix select of the k-th largest value (1-based) of a cube that is spread over the ranks.

    `local_hist(level, prefix)` returns this rank's 2048-bin histogram of the level (numpy uint64 /
    sequence; on the GPU: thz_select_histogram into a zeroed buffer, then a 16 KB download).
    Histograms of tiles add up, so every level costs one all-reduce of 2048 int64 — the only exchange
    step; every rank ends with the same three bins.  Returns (bin0, bin1, bin2); the value is
    binding.host_select_value(*bins)."""
    import numpy as np
    import torch

    from . import binding

    def reduced(level, prefix):
        h = torch.from_numpy(np.ascontiguousarray(local_hist(level, prefix), np.uint64).view(np.int64).copy())
        if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
            if device is not None:
                h = h.to(device)
            dist.all_reduce(h)
            h = h.cpu()
        return h.numpy().view(np.uint64)

    floor = SELECT_FLOOR_BIN
    while True:
        b0, rank_in = binding.host_select_step(reduced(0, floor), k)
        if floor == 0 or b0 != floor:
            break
        floor = 0     # the k-th largest lies in the lump: repeat level 0 in full
    b1, rank_in = binding.host_select_step(reduced(1, b0), rank_in)
    b2, _ = binding.host_select_step(reduced(2, (b0 << 11) | b1)[:1024], rank_in)
    return b0, b1, b2


def voxel_threshold(local_hist, n_local: int, max_instances: int, dist=None, device=None) -> float:
    """effective threshold of instance_from_data for a sharded opacity cube: 0.0 when the whole cube
    has no more than max_instances voxels, else the max_instances-th largest opacity."""
    import torch

    from . import binding

    n = torch.tensor([n_local], dtype=torch.int64)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        if device is not None:
            n = n.to(device)
        dist.all_reduce(n)
        n = n.cpu()
    if int(n.item()) <= max_instances:
        return 0.0
    return binding.host_select_value(*select_kth_largest(local_hist, max_instances, dist, device))
