"""3-D voxel envelope (K15, gui/threed_plot.rs:80-276) on the GPU through the C ABI,
against the oracle.  Tolerances: opacities <= 1e-5 (max-norm; they live in [0, 1]);
selection, counts, ordering, positions and alphas are index work -> bit-exact given the
same opacity cube; colours go through powf(., 2.4) -> 1e-6."""
import numpy as np
import pytest

import oracle_binding as ob
import synth
import thz_image_explorer_amd as pkg
from test_voxel_cpu import envelope_cube

pytestmark = pytest.mark.gpu


def gpu_opacity(eng, cube, cfg):
    nx, ny, nt = cube.shape
    d_in = eng.to_device(cube)
    d_out = eng.empty(cube.shape)
    eng.voxel_opacity(nx * ny, nt, d_in, cfg, d_out)
    out = d_out.download(cube.shape, np.float32)
    return out, d_in, d_out


@pytest.mark.parametrize("shape,radius,contrast,sigma", [((8, 8, 1024), 9, 2.0, 3.0), ((4, 4, 4096), 9, 2.0, 3.0),
                                                         ((3, 3, 1001), 12, 1.0, 5.0), ((3, 3, 300), 20, 2.0, 8.0),
                                                         ((2, 3, 8192), 50, 2.0, 50.0), ((3, 2, 2048), 1, 0.37, 0.1),
                                                         ((2, 2, 5), 9, 2.0, 3.0), ((5, 5, 256), 0, 2.0, 3.0)])
def test_opacity_vs_oracle(engine, shape, radius, contrast, sigma):
    cube = envelope_cube(*shape)
    cfg = pkg.voxel_cfg_default()
    cfg.radius, cfg.contrast, cfg.sigma = radius, contrast, sigma
    if contrast != 2.0:
        cfg.opacity_threshold = 1e-3
    out, d_in, d_out = gpu_opacity(engine, cube, cfg)
    ref = ob.voxel_opacity(cube, sigma, radius, contrast, cfg.opacity_threshold)
    assert np.abs(out - ref).max() < 1e-5
    nt = shape[2]
    live = ref.reshape(-1, nt).max(axis=1) > 0
    assert np.array_equal(out.reshape(-1, nt).max(axis=1) > 0, live)
    if live.any():
        assert np.all(out.reshape(-1, nt)[live].max(axis=1) == 1.0)
        assert np.all(out.reshape(-1, nt)[live].min(axis=1) == 0.0)
    d_in.free(); d_out.free()


def test_opacity_argument_checks(engine):
    cfg = pkg.voxel_cfg_default()
    d = engine.empty((2, 9000))
    with pytest.raises(pkg.ThzError):
        engine.voxel_opacity(2, 9000, d, cfg, engine.empty((2, 9000)))   # nt > 8192
    with pytest.raises(pkg.ThzError):
        engine.voxel_opacity(2, 64, d, cfg, d)                            # in place
    d.free()


@pytest.mark.parametrize("n", [1, 5, 1000, 262147, 3_000_001])
def test_kth_largest_exact(engine, n):
    rng = np.random.default_rng(n)
    v = rng.random(n).astype(np.float32)
    v[rng.random(n) < 0.4] = 0.0
    v[rng.random(n) < 0.1] = np.float32(1.0)
    if n > 100:
        v[:4] = [-3.0, -0.0, 7.5, 1e-40]
    d = engine.to_device(v)
    desc = np.sort(v)[::-1]
    for k in sorted(k for k in {1, 2, max(1, n // 1000), max(1, n // 7), max(1, n // 2), n} if k <= n):
        got = engine.kth_largest(d, n, k)
        assert got == desc[k - 1], (n, k)
    with pytest.raises(pkg.ThzError):
        engine.kth_largest(d, n, n + 1)
    # threed_plot.rs:207-214: no selection when everything fits
    assert engine.voxel_threshold(d, n, max_instances=n) == 0.0
    if n > 1:
        assert engine.voxel_threshold(d, n, max_instances=n - 1) == desc[n - 2]
    d.free()


def test_envelope_to_instances_flow(engine):
    nx, ny, nt = 12, 10, 1024
    cube = envelope_cube(nx, ny, nt, dead_rows=2)
    time = synth.make_time(nt)
    span = float(time[-1] - time[0])
    cfg = pkg.voxel_cfg_default()
    out, d_in, d_op = gpu_opacity(engine, cube, cfg)
    max_inst = 5000
    thr = engine.voxel_threshold(d_op, out.size, max_inst)
    assert thr == ob.voxel_threshold(out, max_inst)           # same cube in -> same element out
    assert abs(thr - ob.voxel_threshold(ob.voxel_opacity(cube), max_inst)) < 1e-5
    ref, rdims = ob.voxel_instances(out, thr, span, 1, (nx, ny, nt))
    cap = len(ref) + 16
    d_inst = engine.alloc(cap * pkg.VOXEL_INSTANCE.itemsize).zero()
    count, dims = engine.voxel_instances(d_op, nx, ny, nt, thr, span, 1, (nx, ny, nt), d_inst, cap)
    assert count == len(ref) >= max_inst
    assert dims == rdims
    got = d_inst.download((cap,), pkg.VOXEL_INSTANCE)
    assert np.array_equal(got[:count]["position"], ref["position"])
    assert np.array_equal(got[:count]["scale"], ref["scale"])
    assert np.array_equal(got[:count]["color"][:, 3], ref["color"][:, 3])
    assert np.abs(got[:count]["color"][:, :3] - ref["color"][:, :3]).max() < 1e-6
    assert np.all(got[count:]["scale"] == 0)
    # x-slab tiles (multi-GPU layout): the two halves concatenate to the whole list
    half = nx // 2
    parts = []
    for x0, gw in ((0, half), (half, nx - half)):
        d_t = engine.alloc(cap * pkg.VOXEL_INSTANCE.itemsize).zero()
        c, _ = engine.voxel_instances(d_op.ptr + x0 * ny * nt * 4, gw, ny, nt, thr, span, 1, (nx, ny, nt), d_t, cap,
                                      x0=x0, gw_total=nx)
        parts.append(d_t.download((cap,), pkg.VOXEL_INSTANCE)[:c])
        d_t.free()
    both = np.concatenate(parts)
    assert len(both) == count and np.array_equal(both["position"], ref["position"])
    # scaled-down cube (scaling 2 of a 24x20 scan): spacing from the original dimensions
    ref2, _ = ob.voxel_instances(out, thr, span, 2, (2 * nx, 2 * ny, nt))
    c2, _ = engine.voxel_instances(d_op, nx, ny, nt, thr, span, 2, (2 * nx, 2 * ny, nt), d_inst, cap)
    got2 = d_inst.download((cap,), pkg.VOXEL_INSTANCE)[:c2]
    assert np.array_equal(got2["position"], ref2["position"]) and np.all(got2["scale"] == 2.0)
    # capacity smaller than the count: count still complete, nothing written past capacity
    d_small = engine.alloc(8 * pkg.VOXEL_INSTANCE.itemsize).zero()
    c3, _ = engine.voxel_instances(d_op, nx, ny, nt, thr, span, 1, (nx, ny, nt), d_small, 5)
    small = d_small.download((8,), pkg.VOXEL_INSTANCE)
    assert c3 == count and np.array_equal(small[:5]["position"], ref[:5]["position"]) and np.all(small[5:]["scale"] == 0)
    for b in (d_in, d_op, d_inst, d_small):
        b.free()


def test_full_size_properties(engine):
    """256 x 256 x 1024 device-synthesised cube (config A): size-independent checks"""
    nx, ny, nt = 256, 256, 1024
    npix = nx * ny
    time = synth.make_time(nt)
    d_time = engine.to_device(time)
    d_cube = engine.empty((npix, nt))
    engine.set_time_axis(time)
    engine.synth_cube(d_cube, npix, 0, d_time)
    # scale to O(1..10) amplitudes like a raw scan so that lines survive the opacity threshold
    d_gain = engine.to_device(np.full(nt, 4.0, np.float32))
    engine.apply_td_window(npix, d_cube, d_gain, d_cube)
    d_op = engine.empty((npix, nt))
    cfg = pkg.voxel_cfg_default()
    engine.voxel_opacity(npix, nt, d_cube, cfg, d_op)
    op = d_op.download((npix, nt), np.float32)
    mx, mn = op.max(axis=1), op.min(axis=1)
    live = mx > 0
    assert 0.2 < live.mean() <= 1.0
    assert np.all(mx[live] == 1.0) and np.all(mn == 0.0)
    # sample of traces against the oracle
    idx = np.arange(0, npix, 997)
    cube_s = d_cube.download((npix, nt), np.float32)[idx]
    assert np.abs(op[idx] - ob.voxel_opacity(cube_s)).max() < 1e-5
    thr = engine.voxel_threshold(d_op, op.size)
    k = pkg.VOXEL_MAX_INSTANCES
    assert thr == np.partition(op.ravel(), op.size - k)[op.size - k]
    n_ge = int((op >= thr).sum())
    d_inst = engine.alloc((n_ge + 1) * pkg.VOXEL_INSTANCE.itemsize)
    count, _ = engine.voxel_instances(d_op, nx, ny, nt, thr, float(time[-1] - time[0]), 1, (nx, ny, nt), d_inst, n_ge + 1)
    assert count == n_ge >= k
    inst = d_inst.download((n_ge,), pkg.VOXEL_INSTANCE)
    assert np.array_equal(inst["color"][:, 3], op[op >= thr])     # x, y, z order == C order of the cube
    for b in (d_time, d_cube, d_gain, d_op, d_inst):
        b.free()


def test_sharded_select_matches_single_gpu(engine):
    """shard.select_kth_largest over two x-slab tiles of one cube (histograms add up; on N GPUs the
    sum is an all-reduce) gives the element thz_kth_largest finds on the whole cube"""
    from thz_image_explorer_amd import shard

    rng = np.random.default_rng(8)
    n = 400_003
    v = rng.random(n).astype(np.float32)
    v[rng.random(n) < 0.5] = 0.0
    d = engine.to_device(v)
    cut = 4 * (n // 8)                       # second tile starts 16-byte aligned
    d_hist = engine.alloc(2048 * 8)

    def local_hist(level, prefix):
        d_hist.zero()
        engine.select_histogram(d.ptr, cut, level, prefix, d_hist)
        engine.select_histogram(d.ptr + 4 * cut, n - cut, level, prefix, d_hist)
        return d_hist.download((2048,), np.uint64)

    for k in (1, 777, 150_000, 250_000):
        bins = shard.select_kth_largest(local_hist, k)
        assert pkg.host_select_value(*bins) == engine.kth_largest(d, n, k) == np.sort(v)[::-1][k - 1]
    d.free(); d_hist.free()
