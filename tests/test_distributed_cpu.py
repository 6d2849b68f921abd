"""N>1 path on the CPU: two gloo ranks shard a cube by x-slabs (the library's rule, thz_host_slab), run the
chain on their slab (the oracle stands in for the GPU here) and go through the exchange steps of
thz_group_session_upload / _recompute with gloo standing in for RCCL:
  upload     all-reduce of the slabs' raw pixel sums
  recompute  C2 all-reduce of the amplitude / phase sums; avg_fft by linearity from the mean trace;
             C1 gather of the image slabs to rank 0
— and must reproduce the single-process result.  (The RCCL calls themselves need >= 2 GPUs: the driver's
multi-GPU bench goes through them; tests/test_gpu_group.py runs the same library code on one device.)"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nx, ny, nt, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import oracle_binding as ob
    import synth
    from thz_image_explorer_amd import shard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import thz_image_explorer_amd as pkg
    x0, nxl = pkg.host_slab(nx, world, rank)
    assert (x0, nxl) == shard.slab(nx, world, rank)
    time, cube = synth.make_cube(nxl, ny, nt, x0=x0, ny_total=ny)
    chain = synth.oracle_chain(time)
    nf = nt // 2 + 1
    # upload: raw pixel sums of the slab, all-reduced (f32, like the device buffers)
    t_raw = torch.from_numpy(cube.reshape(-1, nt).sum(0, dtype=np.float32))
    shard.all_reduce_sums(t_raw, dist)
    # recompute: slab chain, C2 of the amplitude / phase sums, avg_fft from the mean trace, C1 of the image
    res = ob.run_pipeline(cube, time, chain)
    t_sums = torch.from_numpy(np.concatenate([res["amplitudes"].reshape(-1, nf).sum(0, dtype=np.float32),
                                              res["phases"].reshape(-1, nf).sum(0, dtype=np.float32)]))
    shard.all_reduce_sums(t_sums, dist)
    mean_trace = (t_raw.numpy() * np.float32(1.0 / (nx * ny))).reshape(1, 1, nt)
    avg = ob.run_pipeline(mean_trace, time, chain)["fft"].ravel()      # mask * FFT(pre * mean trace)
    img = shard.gather_image(torch.from_numpy(res["img"]), nx, dist)
    if rank == 0:
        q.put((np.concatenate([avg, t_sums.numpy() * np.float32(1.0 / (nx * ny))]), img.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nx", [6, 5])  # even split and a ragged one
def test_two_rank_slab_sharding_matches_single_process(nx):
    import torch.multiprocessing as mp

    import oracle_binding as ob
    import synth

    ny, nt, world = 4, 256, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nx, ny, nt, q)) for r in range(world)]
    for p in procs:
        p.start()
    means, img = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    time, cube = synth.make_cube(nx, ny, nt)
    chain = synth.oracle_chain(time)
    ref = ob.run_pipeline(cube, time, chain)
    nf = nt // 2 + 1
    assert np.array_equal(img, ref["img"])  # slabs are bit-identical to the whole-cube run
    ref_means = np.concatenate([ob.pixel_mean(ref["fft"], 2).ravel(), ob.pixel_mean(ref["amplitudes"], 1),
                                ob.pixel_mean(ref["phases"], 1)])
    # each of the three mean vectors against its own scale: avg_fft by linearity, amplitude / phase means by sums
    for a, b in ((0, 2 * nf), (2 * nf, 3 * nf), (3 * nf, 4 * nf)):
        assert np.abs(means[a:b] - ref_means[a:b]).max() / np.abs(ref_means[a:b]).max() < 1e-5


def test_slab_partition_covers_grid():
    from thz_image_explorer_amd import shard

    for nx in (1, 7, 8, 1024):
        for world in (1, 2, 3, 8):
            rows = [shard.slab(nx, world, r) for r in range(world)]
            assert rows[0][0] == 0
            assert sum(n for _, n in rows) == nx
            for (a, n), (b, _) in zip(rows, rows[1:]):
                assert a + n == b


# ---- voxel threshold of a cube spread over two ranks (shard.voxel_threshold) -----------------
def _float_keys(v):
    b = np.ascontiguousarray(v, np.float32).view(np.uint32)
    return np.where(b & 0x80000000, ~b, b | 0x80000000).astype(np.uint32)


def _numpy_hist(vals):
    """stand-in for thz_select_histogram on a CPU-only box: same bins, same floor rule"""
    keys = _float_keys(vals)

    def hist(level, prefix):
        if level == 0:
            return np.bincount(np.maximum(keys >> 21, prefix), minlength=2048).astype(np.uint64)
        if level == 1:
            return np.bincount((keys[(keys >> 21) == prefix] >> 10) & 2047, minlength=2048).astype(np.uint64)
        return np.bincount(keys[(keys >> 10) == prefix] & 1023, minlength=2048).astype(np.uint64)

    return hist


def _voxel_worker(rank, world, port, seed, n, k, tiny, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    from thz_image_explorer_amd import shard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    vals = _voxel_values(seed, n, tiny)
    x0, nl = shard.slab(n, world, rank)
    thr = shard.voxel_threshold(_numpy_hist(vals[x0:x0 + nl]), nl, k, dist)
    q.put((rank, thr))
    dist.barrier()
    dist.destroy_process_group()


def _voxel_values(seed, n, tiny):
    rng = np.random.default_rng(seed)
    v = rng.random(n).astype(np.float32)
    v[rng.random(n) < 0.5] = 0.0
    v[rng.random(n) < 0.05] = np.float32(1.0)
    if tiny:
        v *= np.float32(1e-5)      # the k-th largest falls below the level-0 floor -> second level-0 round
    return v


@pytest.mark.parametrize("n,k,tiny", [(50_001, 2_000, False), (50_001, 30_000, False), (20_000, 500, True),
                                      (1_000, 5_000, False)])
def test_two_rank_voxel_threshold(n, k, tiny):
    import torch.multiprocessing as mp

    world, seed = 2, 11
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_voxel_worker, args=(r, world, port, seed, n, k, tiny, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    vals = _voxel_values(seed, n, tiny)
    want = 0.0 if n <= k else float(np.sort(vals)[::-1][k - 1])
    assert got[0] == got[1] == want
