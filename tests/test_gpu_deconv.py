"""Frequency-dependent Richardson–Lucy deconvolution (K12) on the GPU against
the oracle's restatement of src/filters/deconvolution.rs:766-1041.  The
reference never runs this math in its own tests (they stop at the 16x16 guard),
so the oracle is the only checker; both its kernel-size branches are covered."""
import os

import numpy as np
import pytest

import oracle_binding as ob
import synth
import thz_image_explorer_amd as pkg

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _bar_target_cube(nx, ny, nt):
    """stand-in for the absent resolution_target_sample.thzimg:
    synthetic traces x a bar-target transmission mask"""
    time, cube = synth.make_cube(nx, ny, nt)
    xx, yy = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    mask = 0.35 + 0.65 * (((xx // 3) % 2 == 0) & (yy > ny // 4) & (yy < 3 * ny // 4))
    mask = mask + 0.3 * (((yy // 5) % 2 == 0) & (xx > nx // 2))
    return time, (cube * mask[..., None].astype(np.float32)).astype(np.float32)


@pytest.mark.parametrize("case", [
    dict(nx=32, ny=32, nt=256, dx=0.5, dy=0.5, n_iter=20, n_filters=6, f0=0.4, f1=3.0, mode=0),
    dict(nx=48, ny=40, nt=128, dx=1.0, dy=1.0, n_iter=12, n_filters=4, f0=0.25, f1=2.0, mode=1),
    # the FIR transform length M = next_pow2(nt + 498) picks the kernels: 1024 above, 2048 and 4096 (both
    # on the register-resident core) and 8192 (generic LDS transform) here
    dict(nx=20, ny=18, nt=1001, dx=0.5, dy=0.5, n_iter=6, n_filters=5, f0=0.4, f1=3.0, mode=0),
    dict(nx=18, ny=20, nt=2000, dx=0.5, dy=0.5, n_iter=6, n_filters=4, f0=0.4, f1=3.0, mode=0),
    dict(nx=16, ny=17, nt=4000, dx=0.5, dy=0.5, n_iter=4, n_filters=3, f0=0.4, f1=3.0, mode=0),
    # many iterations on wide kernels (the reference's FFT-convolution branch; split, fused sums on the device)
    dict(nx=48, ny=40, nt=128, dx=1.0, dy=1.0, n_iter=200, n_filters=4, f0=0.25, f1=2.0, mode=1, loose=True),
])
def test_deconvolution_vs_oracle(engine, case):
    z = np.load(os.path.join(GOLD, "psf_sample.npz"))
    psf, opsf = pkg.psf_from_npz(z), ob.psf_from_npz(z)
    nx, ny, nt = case["nx"], case["ny"], case["nt"]
    time, cube = _bar_target_cube(nx, ny, nt)
    cfg = pkg.DeconvCfg(case["n_iter"], case["n_filters"], case["f0"], case["f1"], 0.5)
    # the case really exercises the intended convolution branch
    sizes = [pkg.host_band_psf(psf, f, case["dx"], case["dy"], nx, ny).size
             for f in pkg.host_filter_bank(time, cfg)[1]]
    assert (max(sizes) > 256) == (case["mode"] == 1)
    rc, oref, oimg, ogains, oniter = ob.deconvolution(cube, time, case["dx"], case["dy"], opsf, case["n_iter"],
                                                      case["n_filters"], case["f0"], case["f1"], 0.5)
    assert rc == 0 and oniter.max() > 1
    e = engine
    e.set_time_axis(time)
    d_in = e.to_device(cube); d_out = e.empty((nx * ny, nt)); d_img = e.empty((nx * ny,))
    d_g = e.empty((case["n_filters"], nx * ny))
    status = e.deconvolve(psf, cfg, nx, ny, case["dx"], case["dy"], d_in, d_out, d_img, d_g)
    assert status == 0
    out = d_out.download((nx, ny, nt), np.float32)
    img = d_img.download((nx, ny), np.float32)
    gains = d_g.download((case["n_filters"], nx, ny), np.float32)
    assert np.isfinite(out).all()
    # Error budget (scripts/gpu_deconv_error_budget.py, profiles/r02_deconv_error_budget.txt): the
    # device's fp32 FIR against the reference's Complex<f64> one costs 3e-7 on the cube and a few 1e-6 on the gains,
    # and Richardson-Lucy does not amplify it (measured 2.5e-7 ... 8e-7 / 3e-7 ... 2.3e-6 / 5e-7 ... 2.2e-6 for
    # cube / gains / image on these cases; 5.6e-6 / 8.7e-6 / 7.1e-6 after 200 iterations on wide kernels)
    bar = 3.0 if case.get("loose") else 1.0
    assert np.abs(out - oref).max() / np.abs(oref).max() < 1e-5 * bar
    assert np.abs(gains - ogains).max() / np.abs(ogains).max() < 1e-5 * bar
    assert np.abs(img - oimg).max() / oimg.max() < 1e-5 * bar
    # the filter does something: it is not the identity
    assert np.abs(out - cube).max() / np.abs(cube).max() > 1e-2
    for b in (d_in, d_out, d_img, d_g):
        b.free()


def test_deconvolution_guards_return_input(engine):
    """deconvolution.rs:1139-1177 (the reference's own test): 2x2 < MIN_IMAGE_SIZE ->
    input returned unchanged, shape preserved; same for an empty PSF"""
    z = np.load(os.path.join(GOLD, "psf_sample.npz"))
    psf = pkg.psf_from_npz(z)
    nt = 64
    time = synth.make_time(nt)
    engine.set_time_axis(time)
    data = synth.make_traces(np.arange(4), nt).reshape(2, 2, nt)
    cfg = pkg.DeconvCfg(500, 25, 0.1, 10.0, 0.5)
    d_in = engine.to_device(data); d_out = engine.empty((4, nt)); d_img = engine.empty((4,))
    assert engine.deconvolve(psf, cfg, 2, 2, 1.0, 1.0, d_in, d_out, d_img) == 1  # THZ_SKIPPED
    assert np.array_equal(d_out.download((2, 2, nt), np.float32), data)
    assert np.allclose(d_img.download((4,), np.float32), (data.reshape(4, nt) ** 2).sum(1), rtol=1e-5)
    empty = pkg.Psf()
    big = synth.make_traces(np.arange(20 * 20), nt).reshape(20, 20, nt)
    d2 = engine.to_device(big); o2 = engine.empty((400, nt))
    assert engine.deconvolve(empty, cfg, 20, 20, 1.0, 1.0, d2, o2) == 1
    assert np.array_equal(o2.download((20, 20, nt), np.float32), big)
    # PSF wider than the image (0.1 THz beam ~ 9 mm on a 20 x 0.5 mm image)
    assert engine.deconvolve(psf, cfg, 20, 20, 0.5, 0.5, d2, o2) == 1
    for b in (d_in, d_out, d_img, d2, o2):
        b.free()


def test_unit_gain_bank_reconstructs_input(engine):
    """the FIR bank sums to a unit impulse, so with zero RL iterations... n_iter >= 1
    always; instead check linearity: deconvolve(a*x) = a*deconvolve(x) (gains are scale-free)"""
    z = np.load(os.path.join(GOLD, "psf_sample.npz"))
    psf = pkg.psf_from_npz(z)
    nx = ny = 24
    nt = 128
    time, cube = _bar_target_cube(nx, ny, nt)
    engine.set_time_axis(time)
    cfg = pkg.DeconvCfg(5, 3, 0.5, 2.0, 0.5)
    outs = []
    for scale in (1.0, 4.0):
        d_in = engine.to_device((cube * np.float32(scale)).astype(np.float32)); d_out = engine.empty((nx * ny, nt))
        assert engine.deconvolve(psf, cfg, nx, ny, 0.5, 0.5, d_in, d_out) == 0
        outs.append(d_out.download((nx, ny, nt), np.float32))
        d_in.free(); d_out.free()
    assert np.abs(outs[1] - 4.0 * outs[0]).max() / np.abs(outs[1]).max() < 1e-5


def test_band_parallel_partials_add_up(engine):
    """config 4 / SURVEY §8e: bands split over ranks, outputs all-reduced.  Two
    band subsets computed one after the other must add up to the full result."""
    z = np.load(os.path.join(GOLD, "psf_sample.npz"))
    psf = pkg.psf_from_npz(z)
    nx = ny = 32
    nt = 256
    time, cube = _bar_target_cube(nx, ny, nt)
    engine.set_time_axis(time)
    d_in = engine.to_device(cube)
    outs = {}
    for name, (b0, b1) in {"all": (0, 0), "lo": (0, 3), "hi": (3, 6)}.items():
        cfg = pkg.DeconvCfg(20, 6, 0.4, 3.0, 0.5, b0, b1)
        d_out = engine.empty((nx * ny, nt))
        assert engine.deconvolve(psf, cfg, nx, ny, 0.5, 0.5, d_in, d_out) == 0
        outs[name] = d_out.download((nx, ny, nt), np.float32)
        d_out.free()
    d_in.free()
    s = outs["lo"] + outs["hi"]
    assert np.abs(s - outs["all"]).max() / np.abs(outs["all"]).max() < 1e-5
    with pytest.raises(pkg.ThzError):
        engine.deconvolve(psf, pkg.DeconvCfg(20, 6, 0.4, 3.0, 0.5, 4, 2), nx, ny, 0.5, 0.5, 0, 0)
    # the partial outputs add up on the pass-through paths too: a guard (PSF wider than a 20 x 20
    # image) makes the rank that owns band 0 copy the input through, every other rank contribute zeros; a rank
    # with no band at all (more ranks than bands) contributes zeros and succeeds
    big = synth.make_traces(np.arange(20 * 20), nt).reshape(20, 20, nt)
    d2 = engine.to_device(big)
    total = np.zeros_like(big)
    for b0, b1 in ((0, 2), (2, 4), (4, 6)):
        o2 = engine.empty((400, nt)).zero()
        assert engine.deconvolve(psf, pkg.DeconvCfg(20, 6, 0.1, 10.0, 0.5, b0, b1), 20, 20, 0.5, 0.5, d2, o2) == 1
        part = o2.download((20, 20, nt), np.float32)
        assert np.array_equal(part, big if b0 == 0 else np.zeros_like(big))
        total += part
        o2.free()
    assert np.array_equal(total, big)
    d3 = engine.to_device(cube); o3 = engine.empty((nx * ny, nt))
    assert engine.deconvolve(psf, pkg.DeconvCfg(20, 6, 0.4, 3.0, 0.5, 3, 3), nx, ny, 0.5, 0.5, d3, o3) == 0
    assert not o3.download((nx, ny, nt), np.float32).any()
    for b in (d2, d3, o3):
        b.free()


def test_deconvolution_abort_and_progress(engine):
    """abort flag (filters/filter.rs: Arc<AtomicBool>) polled between iteration batches: an aborted run
    returns THZ_ERR_ABORTED with the input copied through; a completed run reports progress 1.0"""
    import ctypes
    z = np.load(os.path.join(GOLD, "psf_sample.npz"))
    psf = pkg.psf_from_npz(z)
    nx, ny, nt = 32, 32, 128
    time, cube = _bar_target_cube(nx, ny, nt)
    engine.set_time_axis(time)
    cfg = pkg.DeconvCfg(80, 5, 0.3, 3.0, 0.5)     # > one graph batch of iterations
    d_in = engine.to_device(cube); d_out = engine.empty((nx * ny, nt)).zero(); d_img = engine.empty((nx * ny,))
    abort, progress = ctypes.c_int(1), ctypes.c_float(-1.0)
    with pytest.raises(pkg.ThzError) as e:
        engine.deconvolve(psf, cfg, nx, ny, 0.5, 0.5, d_in, d_out, d_img, abort=abort, progress=progress)
    assert e.value.code == -5
    assert np.array_equal(d_out.download((nx, ny, nt), np.float32), cube)
    abort.value = 0
    assert engine.deconvolve(psf, cfg, nx, ny, 0.5, 0.5, d_in, d_out, d_img, abort=abort, progress=progress) == 0
    assert progress.value == 1.0
    assert np.abs(d_out.download((nx, ny, nt), np.float32) - cube).max() > 0
    for b in (d_in, d_out, d_img):
        b.free()


def test_untiled_fallback_matches_tiled(engine, monkeypatch):
    """PSFs too wide for an LDS tile take k_rl_step (every tap from L2); forced here with the developer knob:
    narrow kernels in the reference's own order of every sum (THZ_RL_NARROW_EXACT: k_rl_step_tiled<false>, the
    reference's direct sums) must agree bit for bit, wide ones within rounding"""
    monkeypatch.setenv("THZ_RL_NARROW_EXACT", "1")
    z = np.load(os.path.join(GOLD, "psf_sample.npz"))
    psf = pkg.psf_from_npz(z)
    for case in (dict(nx=32, ny=32, nt=256, d=0.5, cfg=pkg.DeconvCfg(10, 4, 0.8, 3.0, 0.5), exact=True),
                 dict(nx=48, ny=40, nt=128, d=1.0, cfg=pkg.DeconvCfg(8, 4, 0.25, 2.0, 0.5), exact=False)):
        nx, ny, nt = case["nx"], case["ny"], case["nt"]
        time, cube = _bar_target_cube(nx, ny, nt)
        sizes = [pkg.host_band_psf(psf, f, case["d"], case["d"], nx, ny).size
                 for f in pkg.host_filter_bank(time, case["cfg"])[1]]
        assert (max(sizes) <= 256) == case["exact"]
        engine.set_time_axis(time)
        d_in = engine.to_device(cube); d_out = engine.empty((nx * ny, nt))
        outs = []
        for knob in (None, "1"):
            if knob:
                monkeypatch.setenv("THZ_NO_TILE", knob)
            else:
                monkeypatch.delenv("THZ_NO_TILE", raising=False)
            assert engine.deconvolve(psf, case["cfg"], nx, ny, case["d"], case["d"], d_in, d_out) == 0
            outs.append(d_out.download((nx, ny, nt), np.float32))
        monkeypatch.delenv("THZ_NO_TILE", raising=False)
        if case["exact"]:
            assert np.array_equal(outs[0], outs[1])
        else:
            assert np.abs(outs[0] - outs[1]).max() / np.abs(outs[1]).max() < 1e-5
        d_in.free(); d_out.free()


def test_separable_wide_kernels_match_the_2d_sums(engine, monkeypatch):
    """every band PSF is an outer product of two profiles, so the wide kernels run as a pass along the rows and a
    pass down the columns (k_rl_step_sep); THZ_RL_NO_SEPARABLE=1 keeps the 2-D sums of round 1 — the two must
    agree within rounding, and the host's 2-D array must BE the outer product the device factors it into"""
    z = np.load(os.path.join(GOLD, "psf_sample.npz"))
    psf = pkg.psf_from_npz(z)
    nx, ny, nt, d = 48, 40, 128, 1.0
    cfg = pkg.DeconvCfg(30, 4, 0.25, 2.0, 0.5)
    time, cube = _bar_target_cube(nx, ny, nt)
    wide = 0
    for f in pkg.host_filter_bank(time, cfg)[1]:
        k = pkg.host_band_psf(psf, f, d, d, nx, ny)
        wide += k.size > 256
        c = np.unravel_index(np.argmax(k), k.shape)   # normalised profiles: the peak row / column are the factors
        assert np.abs(np.outer(k[:, c[1]], k[c[0], :]) / k[c] - k).max() <= 4e-7 * k.max()
    assert wide >= 1
    engine.set_time_axis(time)
    d_in = engine.to_device(cube); d_out = engine.empty((nx * ny, nt)); d_g = engine.empty((4, nx * ny))
    res = []
    for knob in (None, "1"):
        if knob:
            monkeypatch.setenv("THZ_RL_NO_SEPARABLE", knob)
        else:
            monkeypatch.delenv("THZ_RL_NO_SEPARABLE", raising=False)
        assert engine.deconvolve(psf, cfg, nx, ny, d, d, d_in, d_out, None, d_g) == 0
        res.append((d_out.download((nx, ny, nt), np.float32), d_g.download((4, nx, ny), np.float32)))
    monkeypatch.delenv("THZ_RL_NO_SEPARABLE", raising=False)
    assert not np.array_equal(res[0][1], res[1][1])          # two different kernels did run
    assert np.abs(res[0][0] - res[1][0]).max() / np.abs(res[1][0]).max() < 5e-6
    assert np.abs(res[0][1] - res[1][1]).max() / np.abs(res[1][1]).max() < 5e-6
    d_in.free(); d_out.free(); d_g.free()


def test_narrow_kernels_as_two_passes_match_the_reference_order_sums(engine, monkeypatch):
    """kernels of <= 256 taps are the reference's direct sums (deconvolution.rs:432-458, a correlation, m outer / n
    inner); by default they run as the same two 1-D passes as the wide ones (mode 0 of k_rl_step_sep: profiles the other
    way round), THZ_RL_NARROW_EXACT=1 keeps the reference's order of every sum — the two agree within rounding,
    through 40 iterations on every band, and both are within the end-to-end bar of the oracle"""
    z = np.load(os.path.join(GOLD, "psf_sample.npz"))
    psf, opsf = pkg.psf_from_npz(z), ob.psf_from_npz(z)
    nx, ny, nt, d = 40, 36, 256, 0.5
    cfg = pkg.DeconvCfg(40, 5, 0.8, 3.0, 0.5)
    time, cube = _bar_target_cube(nx, ny, nt)
    sizes = [pkg.host_band_psf(psf, f, d, d, nx, ny).shape for f in pkg.host_filter_bank(time, cfg)[1]]
    assert all(a * b <= 256 and a % 2 == 1 and b % 2 == 1 for a, b in sizes) and len(set(sizes)) > 1
    engine.set_time_axis(time)
    d_in = engine.to_device(cube); d_out = engine.empty((nx * ny, nt)); d_g = engine.empty((5, nx * ny))
    res = []
    for knob in (None, "1"):
        if knob:
            monkeypatch.setenv("THZ_RL_NARROW_EXACT", knob)
        else:
            monkeypatch.delenv("THZ_RL_NARROW_EXACT", raising=False)
        assert engine.deconvolve(psf, cfg, nx, ny, d, d, d_in, d_out, None, d_g) == 0
        res.append((d_out.download((nx, ny, nt), np.float32), d_g.download((5, nx, ny), np.float32)))
    monkeypatch.delenv("THZ_RL_NARROW_EXACT", raising=False)
    assert not np.array_equal(res[0][1], res[1][1])          # two different kernels did run
    assert np.abs(res[0][0] - res[1][0]).max() / np.abs(res[1][0]).max() < 2e-6
    assert np.abs(res[0][1] - res[1][1]).max() / np.abs(res[1][1]).max() < 2e-6
    rc, oref, oimg, og, onit = ob.deconvolution(cube, time, d, d, opsf, 40, 5, 0.8, 3.0, 0.5)
    for out, g in res:
        assert np.abs(out - oref).max() / np.abs(oref).max() < 1e-5
        assert np.abs(g - og).max() / np.abs(og).max() < 1e-5
    d_in.free(); d_out.free(); d_g.free()


def test_chain_scheduling_knobs_change_no_bit(engine, monkeypatch):
    """how the Richardson-Lucy chains are laid over the streams — the chains behind the first started late
    (THZ_RL_DELAY), every chain as a replayed graph over its whole tile list (THZ_RL_EXACT=0), plain launches
    (THZ_NO_GRAPH) — is scheduling only: the bands never exchange anything, so every variant returns the same bits"""
    z = np.load(os.path.join(GOLD, "psf_sample.npz"))
    psf = pkg.psf_from_npz(z)
    nx, ny, nt, d = 48, 40, 128, 1.0
    cfg = pkg.DeconvCfg(100, 6, 0.25, 3.0, 0.5)    # more than two batches of 32 iterations, wide and narrow bands
    time, cube = _bar_target_cube(nx, ny, nt)
    sizes = [pkg.host_band_psf(psf, f, d, d, nx, ny).size for f in pkg.host_filter_bank(time, cfg)[1]]
    assert min(sizes) <= 256 < max(sizes)
    engine.set_time_axis(time)
    d_in = engine.to_device(cube); d_out = engine.empty((nx * ny, nt)); d_g = engine.empty((6, nx * ny))
    knobs = ({}, {"THZ_RL_DELAY": "2"}, {"THZ_RL_EXACT": "0"}, {"THZ_NO_GRAPH": "1"}, {"THZ_RL_DELAY": "1", "THZ_RL_NARROW_EXACT": "1"})
    res = []
    for kn in knobs:
        for k, v in kn.items():
            monkeypatch.setenv(k, v)
        assert engine.deconvolve(psf, cfg, nx, ny, d, d, d_in, d_out, None, d_g) == 0
        res.append((d_out.download((nx, ny, nt), np.float32), d_g.download((6, nx, ny), np.float32)))
        for k in kn:
            monkeypatch.delenv(k, raising=False)
    for out, g in res[1:4]:
        assert np.array_equal(out, res[0][0]) and np.array_equal(g, res[0][1])
    # the reference-order narrow sums are another kernel: equal within rounding, not bit for bit
    assert np.abs(res[4][0] - res[0][0]).max() / np.abs(res[0][0]).max() < 2e-6
    d_in.free(); d_out.free(); d_g.free()


def test_repeated_calls_reuse_and_release_device_blocks():
    """the call's device blocks, iteration graphs, FIR bank, transform tables and filter spectra stay in the context for
    the next call (ctx.hpp: dc_pool, dc_graph, dc_bank, dc_plan, dc_spectra): a second call of the same geometry, a
    call of another geometry or another bank in between and a guarded call (which uses none) must all give what a
    fresh context gives, bit for bit"""
    z = np.load(os.path.join(GOLD, "psf_sample.npz"))
    psf = pkg.psf_from_npz(z)
    cases = [dict(nx=32, ny=32, nt=256, d=0.5, cfg=pkg.DeconvCfg(10, 4, 0.8, 3.0, 0.5)),
             dict(nx=48, ny=40, nt=128, d=1.0, cfg=pkg.DeconvCfg(8, 4, 0.25, 2.0, 0.5)),
             dict(nx=12, ny=40, nt=128, d=1.0, cfg=pkg.DeconvCfg(8, 4, 0.25, 2.0, 0.5)),   # guarded: nx < 16
             # the first case's cube and time axis under another bank: the cached FIR bank and filter spectra must go
             dict(nx=32, ny=32, nt=256, d=0.5, cfg=pkg.DeconvCfg(10, 4, 0.6, 2.5, 0.5)),
             dict(nx=32, ny=32, nt=256, d=0.5, cfg=pkg.DeconvCfg(10, 5, 0.8, 3.0, 0.5))]

    def run(eng, c):
        time, cube = _bar_target_cube(c["nx"], c["ny"], c["nt"])
        eng.set_time_axis(time)
        d_in = eng.to_device(cube); d_out = eng.empty((c["nx"] * c["ny"], c["nt"]))
        st = eng.deconvolve(psf, c["cfg"], c["nx"], c["ny"], c["d"], c["d"], d_in, d_out)
        out = d_out.download((c["nx"], c["ny"], c["nt"]), np.float32)
        d_in.free(); d_out.free()
        return st, out

    fresh = []
    for c in cases:
        eng = pkg.Engine(0)
        fresh.append(run(eng, c))
        eng.close()
    eng = pkg.Engine(0)
    for n, i in enumerate((0, 0, 1, 0, 2, 1, 1, 0, 3, 0, 4, 3, 3)):
        st, out = run(eng, cases[i])
        assert st == fresh[i][0] and np.array_equal(out, fresh[i][1]), i
        if n == 5:
            eng.release_scratch()   # thz_release_scratch: the next call allocates again
    eng.close()
