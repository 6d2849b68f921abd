"""Parity of the HIP path (through the C ABI) with the CPU oracle and the
committed golden vectors.  Tolerances: north_star — bit-exact for index/ROI
masks, <= 1e-5 relative (max-norm, per cube) on fp32 spectra and traces."""
import os

import numpy as np
import pytest

import oracle_binding as ob
import synth
import thz_image_explorer_amd as pkg

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-5


def rel(a, b, scale=None):
    scale = np.abs(b).max() if scale is None else scale
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / scale)


def phase_ok(ph, ref, amp_ref):
    """unwrapped phase: equal modulo 2*pi-decision flips on noise bins; no flip
    while the amplitude is above 5 % of the trace maximum"""
    d = ph.astype(np.float64) - ref
    jumps = np.round(d / (2 * np.pi))
    if np.abs(d - 2 * np.pi * jumps).max() > 3e-3:
        return False
    strong = amp_ref > 0.05 * amp_ref.max(axis=-1, keepdims=True)
    flat_j = jumps.reshape(-1, jumps.shape[-1])
    flat_s = strong.reshape(-1, strong.shape[-1])
    for j, s in zip(flat_j, flat_s):
        weak = np.nonzero(~s[3:])[0]
        end = (weak[0] + 3) if weak.size else s.size
        if np.any(j[:end] != 0):
            return False
    return True


def phase_parity(ph, ref_ph, ref_fft, spectrum_tol=TOL):
    """Unwrapped phases against the oracle's, derived from what numpy_unwrap (math_tools.rs:211-240) can and cannot
    pin.  (1) Modulo 2 pi every bin agrees to 3e-3 rad + what the bin's own argument is known to: the running sum
    adds up to nf (~2 049) adjusted differences of magnitude <= pi, each rounded to the sum's ulp (<= 6e-5 at
    |phi| ~ 600 rad) — observed <= 5e-4 — plus the arctangent's own 3e-7.  (2) The multiple of 2 pi may only change at a bin where the reference's own
    wrap decision |arg X[k] - arg X[k-1]| <> pi is closer to its threshold than the two arguments are known: a
    spectrum that is right to spectrum_tol * max|X| (the parity bar) has arg X[k] to within that over |X[k]|.
    Returns (ok, message)."""
    d = np.asarray(ph, np.float64) - np.asarray(ref_ph, np.float64)
    j = np.round(d / (2 * np.pi))
    res = np.abs(d - 2 * np.pi * j)
    X = ref_fft[..., 0].astype(np.float64) + 1j * ref_fft[..., 1].astype(np.float64)
    mag = np.abs(X)
    raw = np.angle(X)
    # how well a spectrum that is right to spectrum_tol * max|X| pins arg X[k]: on a bin deep in the noise floor
    # (water lines, the roll-off) the argument itself is only known to tol * max / |X[k]|
    err = spectrum_tol * mag.max(axis=-1, keepdims=True) / np.maximum(mag, 1e-300) + 1e-6
    over = res > 3e-3 + err
    if over.any():
        idx = np.argwhere(over)[0]
        return False, f"residual {res[tuple(idx)]:.2e} rad modulo 2 pi at {idx.tolist()}, argument known to {err[tuple(idx)]:.2e}"
    dd = np.diff(raw, axis=-1)
    dd = np.abs(np.abs(dd) - np.pi)                               # distance of the decision from its threshold
    slack = err[..., 1:] + err[..., :-1]
    changed = np.diff(j, axis=-1) != 0
    bad = changed & (dd > slack)
    if j[..., 0].any():
        # bin 0 is real: arg is 0 or +-pi; a 2 pi offset there is the sign of a zero imaginary part
        bad0 = (j[..., 0] != 0) & (np.abs(np.abs(raw[..., 0]) - np.pi) > 1e-6)
        if bad0.any():
            return False, f"{int(bad0.sum())} traces offset by 2 pi from bin 0 on"
    if bad.any():
        idx = np.argwhere(bad)[0]
        return False, (f"{int(bad.sum())} wrap decisions differ where the reference's is not borderline, first at {idx.tolist()}: "
                       f"| |dphi| - pi | = {dd[tuple(idx)]:.2e}, slack {slack[tuple(idx)]:.2e}")
    return True, ""


def gpu_fft_stage(eng, cube, w, mask=None, want_data=True):
    nx, ny, nt = cube.shape
    npix, nf = nx * ny, nt // 2 + 1
    d_in = eng.to_device(cube); d_w = eng.to_device(w)
    d_mask = eng.to_device(mask) if mask is not None else None
    d_dat = eng.empty((npix, nt)) if want_data else None
    d_fft = eng.empty((npix, nf, 2)); d_amp = eng.empty((npix, nf)); d_ph = eng.empty((npix, nf))
    eng.fft(npix, d_in, d_w, None, d_dat, d_fft, d_amp, d_ph, d_mask)
    out = dict(fft=d_fft.download((nx, ny, nf, 2), np.float32), amplitudes=d_amp.download((nx, ny, nf), np.float32),
               phases=d_ph.download((nx, ny, nf), np.float32))
    if want_data:
        out["data"] = d_dat.download((nx, ny, nt), np.float32)
    for b in (d_in, d_w, d_mask, d_dat, d_fft, d_amp, d_ph):
        if b is not None:
            b.free()
    return out


@pytest.mark.parametrize("nt", [64, 128, 256, 1024, 2048, 4096, 1001, 1000])
@pytest.mark.parametrize("wtype", [0, 2])
def test_fft_stage_vs_oracle(engine, nt, wtype):
    nx, ny = 3, 7
    time, cube = synth.make_cube(nx, ny, nt)
    engine.set_time_axis(time)
    assert engine.kernel_variant() != ""
    w = pkg.host_fft_window(time, wtype, 1.0, 7.0)
    got = gpu_fft_stage(engine, cube, w)
    ref = ob.fft_stage(cube, time, wtype, 1.0, 7.0)
    assert np.array_equal(got["data"], ref["data"])  # one f32 multiply: bit-exact
    scale = np.abs(ref["fft"]).max()
    assert rel(got["fft"], ref["fft"], scale) < TOL
    assert rel(got["amplitudes"], ref["amplitudes"], scale) < TOL
    assert phase_ok(got["phases"], ref["phases"], ref["amplitudes"])


@pytest.mark.parametrize("nt", [128, 1000, 1001, 1024, 4096])
def test_golden_vectors(engine, nt):
    """GPU vs numpy-fp64 truth (tests/golden/fft_vectors.npz)"""
    v = np.load(os.path.join(GOLD, "fft_vectors.npz"))
    time, raw, freq = v[f"nt{nt}_time"], v[f"nt{nt}_raw"], v[f"nt{nt}_freq"]
    engine.set_time_axis(time)
    assert np.array_equal(engine.frequency(), freq)
    cube = raw.reshape(1, -1, nt)
    for kind in range(5):
        w = pkg.host_fft_window(time, kind, 1.0, 7.0)
        got = gpu_fft_stage(engine, cube, w, want_data=False)
        X = got["fft"][0, ..., 0] + 1j * got["fft"][0, ..., 1]
        ref = v[f"nt{nt}_w{kind}_fft"]
        assert np.abs(X - ref).max() / np.abs(ref).max() < TOL
    w = pkg.host_fft_window(time, 0, 1.0, 7.0)
    mask, lo, up = pkg.host_fd_bandpass(freq, 0.2, 5.0, 0.1)
    assert (lo, up) == tuple(v[f"nt{nt}_fd_idx"])
    got = gpu_fft_stage(engine, cube, w, mask, want_data=False)
    amp_ref = v[f"nt{nt}_w0_amp"]
    assert rel(got["amplitudes"][0], amp_ref * mask) < TOL
    assert phase_ok(got["phases"][0], v[f"nt{nt}_w0_phase"], amp_ref)
    npix = raw.shape[0]
    d_fft = engine.to_device(got["fft"]); d_out = engine.empty((npix, nt))
    engine.ifft(npix, d_fft, None, d_out, None)
    back = d_out.download((npix, nt), np.float32)
    assert rel(back, v[f"nt{nt}_w0_irfft_bp"]) < TOL
    d_fft.free(); d_out.free()


@pytest.mark.parametrize("nt", [256, 1024, 4096, 1001])
def test_stagewise_chain_vs_oracle(engine, nt):
    """scaling(1) -> tilt taper -> Time Band Pass -> fft -> Frequency Band Pass ->
    ifft -> Time Band Pass, one C-ABI call per stage like data_thread.rs:1108-1191"""
    nx, ny = 4, 6
    time, cube = synth.make_cube(nx, ny, nt)
    engine.set_time_axis(time)
    chain = synth.default_chain(time)
    npix, nf = nx * ny, nt // 2 + 1
    e = engine
    d = e.to_device(cube)
    d_tilt = e.to_device(chain["w_tilt"]); d_tdb = e.to_device(chain["w_td_before"])
    d_wfft = e.to_device(chain["w_fft"]); d_fd = e.to_device(chain["fd_mask"]); d_post = e.to_device(chain["w_post"])
    s1 = e.empty((npix, nt)); s2 = e.empty((npix, nt)); s3 = e.empty((npix, nt))
    d_fft = e.empty((npix, nf, 2)); d_amp = e.empty((npix, nf)); d_ph = e.empty((npix, nf))
    s4 = e.empty((npix, nt)); s5 = e.empty((npix, nt)); d_img = e.empty((npix,))
    e.apply_td_window(npix, d, d_tilt, s1)
    e.apply_td_window(npix, s1, d_tdb, s2)
    e.fft(npix, s2, d_wfft, None, s3, d_fft, d_amp, d_ph, None)
    fft_unmasked = d_fft.download((nx, ny, nf, 2), np.float32)
    e.apply_fd_mask(npix, d_fft, d_amp, d_fd)
    e.ifft(npix, d_fft, None, s4, None)
    e.apply_td_window(npix, s4, d_post, s5)
    e.intensity(npix, s5, d_img)
    # oracle, stage by stage
    o1 = cube * chain["w_tilt"]
    o2, _, _ = ob.td_bandpass(o1, time, float(time[0]), float(time[-1]), 2.0)
    assert np.array_equal(s2.download((nx, ny, nt), np.float32), o2)
    st = ob.fft_stage(o2, time, 0, 1.0, 7.0)
    assert np.array_equal(s3.download((nx, ny, nt), np.float32), st["data"])
    scale = np.abs(st["fft"]).max()
    assert rel(fft_unmasked, st["fft"], scale) < TOL
    of, oa = ob.fd_bandpass(st["fft"], st["amplitudes"], chain["frequency"], 0.2, 5.0, 0.1)
    assert rel(d_fft.download((nx, ny, nf, 2), np.float32), of, scale) < TOL
    assert rel(d_amp.download((nx, ny, nf), np.float32), oa, scale) < TOL
    assert phase_ok(d_ph.download((nx, ny, nf), np.float32), st["phases"], st["amplitudes"])
    ot, nerr = ob.ifft_stage(of, nt)
    assert nerr == 0
    assert rel(s4.download((nx, ny, nt), np.float32), ot) < TOL
    o5, _, _ = ob.td_bandpass(ot, time, float(time[0]), float(time[-1]), 0.1)
    assert rel(s5.download((nx, ny, nt), np.float32), o5) < TOL
    assert rel(d_img.download((nx, ny), np.float32), ob.intensity(o5)) < TOL
    for b in (d, d_tilt, d_tdb, d_wfft, d_fd, d_post, s1, s2, s3, d_fft, d_amp, d_ph, s4, s5, d_img):
        b.free()


@pytest.mark.parametrize("shape", [(8, 16, 1024), (2, 3, 4096), (3, 7, 2048), (5, 5, 256), (4, 4, 1001), (1, 1, 128)])
def test_fused_pipeline_vs_oracle(engine, shape):
    nx, ny, nt = shape
    time, cube = synth.make_cube(nx, ny, nt)
    engine.set_time_axis(time)
    # the product computes its own multiplier vectors, the oracle its own: neither side's inputs come from the other
    got = synth.run_gpu_pipeline(engine, cube, synth.default_chain(time))
    chain = synth.oracle_chain(time)
    ref = ob.run_pipeline(cube, time, chain)
    scale = np.abs(ref["fft"]).max()
    assert rel(got["fft"], ref["fft"], scale) < TOL
    assert rel(got["amplitudes"], ref["amplitudes"], scale) < TOL
    assert rel(got["data"], ref["data"]) < TOL
    assert rel(got["img"], ref["img"]) < TOL
    # phases are taken before the band-pass; compare against the un-masked amplitude
    st = ob.fft_stage(cube * chain["w_tilt"] * chain["w_td_before"], time, 0, 1.0, 7.0)
    assert phase_ok(got["phases"], ref["phases"], st["amplitudes"])


def test_knife_edge_real_traces(engine):
    """16 real traces, Nt = 1001 (Bluestein path), default chain vs oracle and numpy fp64"""
    k = np.load(os.path.join(GOLD, "knife_edge.npz"))
    time = k["time"]
    cube = ob.subtract_bias(k["traces"].reshape(4, 4, 1001))
    engine.set_time_axis(time)
    assert engine.kernel_variant().startswith("p-mixed-radix-7x11x13")   # the length of real scans has its own kernel
    got = synth.run_gpu_pipeline(engine, cube, synth.default_chain(time))
    chain = synth.oracle_chain(time)
    ref = ob.run_pipeline(cube, time, chain)
    assert rel(got["fft"], ref["fft"], np.abs(ref["fft"]).max()) < TOL
    assert rel(got["data"], ref["data"]) < TOL
    X = np.fft.rfft(cube.astype(np.float64) * chain["w_pre"], axis=-1) * chain["fd_mask"]
    G = got["fft"][..., 0] + 1j * got["fft"][..., 1]
    assert np.abs(G - X).max() / np.abs(X).max() < TOL


# ---- the reference's unit tests, through the C ABI --------------------------
def test_reference_fft_roundtrip(engine):
    """math_tools.rs:843-897"""
    u = np.load(os.path.join(GOLD, "unit_signals.npz"))
    sig, time = u["roundtrip_signal"], u["roundtrip_time"]
    n = sig.size
    engine.set_time_axis(time)
    w = pkg.host_fft_window(time, 0, 0.0, 0.0)
    assert np.all(w == 1.0)
    got = gpu_fft_stage(engine, sig.reshape(1, 1, n), w)
    d_fft = engine.to_device(got["fft"]); d_out = engine.empty((1, n))
    engine.ifft(1, d_fft, None, d_out, None)
    back = d_out.download((n,), np.float32)
    assert np.abs(back - got["data"].ravel()).max() <= 1e-4
    X = got["fft"][0, 0, :, 0] + 1j * got["fft"][0, 0, :, 1]
    assert np.abs(X - u["roundtrip_fft"]).max() / np.abs(u["roundtrip_fft"]).max() < TOL
    d_fft.free(); d_out.free()


def test_reference_fd_bandpass_exact_zeros(engine):
    """band_pass_fd.rs:475-567"""
    u = np.load(os.path.join(GOLD, "unit_signals.npz"))
    sig, freq = u["fd_signal"], u["fd_freq"]
    n, k = sig.size, 9
    time = np.linspace(0.0, 1.0, n, dtype=np.float32)
    engine.set_time_axis(time)
    mask, lower, upper = pkg.host_fd_bandpass(freq, float(freq[k - 2]), float(freq[k + 2]), 0.0)
    got = gpu_fft_stage(engine, sig.reshape(1, 1, n), np.ones(n, np.float32))
    d_fft = engine.to_device(got["fft"]); d_amp = engine.to_device(got["amplitudes"]); d_m = engine.to_device(mask)
    engine.apply_fd_mask(1, d_fft, d_amp, d_m)
    amp = d_amp.download((n // 2 + 1,), np.float32)
    fft = d_fft.download((n // 2 + 1, 2), np.float32)
    assert np.all(amp[:lower] == 0.0) and np.all(amp[upper:] == 0.0)
    assert np.all(fft[:lower] == 0.0) and np.all(fft[upper:] == 0.0)
    assert amp[lower:upper].sum() > 0
    for b in (d_fft, d_amp, d_m):
        b.free()


def test_reference_td_bandpass_exact_zeros(engine):
    """band_pass_td_before_fft.rs:390-443"""
    u = np.load(os.path.join(GOLD, "unit_signals.npz"))
    time, sig = u["td_time"], u["td_signal"]
    n = sig.size
    engine.set_time_axis(time)
    w, lo, hi, lower, upper = pkg.host_td_bandpass(time, 0.25, 0.55, 0.0)
    assert (lower, upper) == tuple(u["td_idx"])
    d = engine.to_device(sig); d_w = engine.to_device(w); d_o = engine.empty((n,))
    engine.apply_td_window(1, d, d_w, d_o)
    out = d_o.download((n,), np.float32)
    assert np.all(out[:lower] == 0.0) and np.all(out[upper:] == 0.0)
    assert np.array_equal(out[lower:upper], sig[lower:upper])
    for b in (d, d_w, d_o):
        b.free()


# ---- reductions, ROI, scaling: bit-exact ---------------------------------
def test_pixel_mean_bit_exact(engine):
    rng = np.random.default_rng(0)
    nx, ny, nf = 9, 13, 129
    engine.set_time_axis(synth.make_time(256))
    a = rng.standard_normal((nx, ny, nf)).astype(np.float32)
    c = rng.standard_normal((nx, ny, nf, 2)).astype(np.float32)
    for arr, ncomp in ((a, 1), (c, 2)):
        d = engine.to_device(arr); o = engine.empty((nf * ncomp,))
        engine.pixel_mean(nx, ny, nf, ncomp, d, o)
        got = o.download((nf * ncomp,), np.float32)
        assert np.array_equal(got, ob.pixel_mean(arr, ncomp).ravel())
        s = engine.empty((nf * ncomp,))
        engine.pixel_sum(nx * ny, nf, ncomp, d, s)
        tot = s.download((nf * ncomp,), np.float32)
        assert np.allclose(tot, arr.reshape(nx * ny, -1).astype(np.float64).sum(0), rtol=1e-5, atol=1e-4)
        d.free(); o.free(); s.free()


def test_roi_masks_and_means_bit_exact(engine):
    g = np.load(os.path.join(GOLD, "roi_masks.npz"))
    names = sorted({k[: -len("_poly")] for k in g.files if k.endswith("_poly")})
    rng = np.random.default_rng(2)
    engine.set_time_axis(synth.make_time(64))
    for (s0, s1) in ((32, 32), (129, 257)):
        data = rng.standard_normal((s0, s1, 33)).astype(np.float32)
        d_data = engine.to_device(data)
        for name in names:
            poly = g[name + "_poly"]
            for scaling in (1, 2):
                d_mask = engine.empty((s0, s1), np.uint8)
                engine.roi_mask(poly, scaling, s0, s1, d_mask)
                mask = d_mask.download((s0, s1), np.uint8)
                assert np.array_equal(mask, g[f"{name}_{s0}x{s1}_s{scaling}_mask"]), (name, s0, scaling)
                d_out = engine.empty((33,)); d_cnt = engine.empty((1,), np.uint32)
                engine.roi_mean(d_data, s0, s1, 33, d_mask, d_out, d_cnt)
                got = d_out.download((33,), np.float32)
                assert np.array_equal(got, ob.average_polygon_roi(data, poly, scaling)), (name, s0, scaling)
                assert int(d_cnt.download((1,), np.uint32)[0]) == int(mask.sum())
                d_mask.free(); d_out.free(); d_cnt.free()
        d_data.free()


def test_roi_empty_mask_gives_zeros(engine):
    engine.set_time_axis(synth.make_time(64))
    data = np.ones((8, 8, 5), np.float32)
    d = engine.to_device(data); m = engine.empty((8, 8), np.uint8); o = engine.to_device(np.full(5, 7, np.float32))
    engine.roi_mask(np.array([[2, 2], [2, 2], [2, 2]], np.uint64), 1, 8, 8, m)
    engine.roi_mean(d, 8, 8, 5, m, o)
    assert np.all(o.download((5,), np.float32) == 0)
    for b in (d, m, o):
        b.free()


@pytest.mark.parametrize("s", [2, 3])
def test_scaling_bit_exact(engine, s):
    rng = np.random.default_rng(4)
    a = rng.standard_normal((7, 9, 40)).astype(np.float32)
    c = rng.standard_normal((7, 9, 21, 2)).astype(np.float32)
    for arr, ncomp, ln in ((a, 1, 40), (c, 2, 21)):
        d = engine.to_device(arr); o = engine.empty((7 // s, 9 // s, ln * ncomp))
        engine.scale3d(d, 7, 9, ln, ncomp, s, o)
        got = o.download(ob.scale3d(arr, s, ncomp).shape, np.float32)
        assert np.array_equal(got, ob.scale3d(arr, s, ncomp))
        d.free(); o.free()


def test_bias_and_intensity(engine):
    nt = 1024
    time = synth.make_time(nt)
    engine.set_time_axis(time)
    raw = synth.make_traces(np.arange(40), nt, subtract_bias=False)
    d = engine.to_device(raw); img = engine.empty((40,))
    engine.subtract_bias(40, d, img)
    got = d.download((40, nt), np.float32)
    ref = ob.subtract_bias(raw)
    assert np.array_equal(got, ref)
    assert rel(img.download((40,), np.float32), ob.intensity(ref)) < TOL
    d.free(); img.free()


def test_device_synth_matches_host(engine):
    nt = 1024
    time = synth.make_time(nt)
    engine.set_time_axis(time)
    d_t = engine.to_device(time); d = engine.empty((50, nt))
    engine.synth_cube(d, 50, 12345, d_t)
    got = d.download((50, nt), np.float32)
    ref = synth.make_traces(np.arange(50) + 12345, nt)
    assert np.abs(got - ref).max() < 1e-5  # expf/logf/sincos ulps, amplitudes ~1
    d_t.free(); d.free()


# ---- edge cases ------------------------------------------------------------
def test_edge_cases(engine):
    e = engine
    with pytest.raises(pkg.ThzError) as ei:
        e.set_time_axis(np.arange(70000, dtype=np.float32))  # above 65536 samples (up to there every length has a plan)
    assert ei.value.code == -2
    with pytest.raises(pkg.ThzError):
        e.set_time_axis(np.zeros(1, np.float32))
    e.set_time_axis(synth.make_time(256))
    with pytest.raises(pkg.ThzError) as ei:
        e.fft(4, None)
    assert ei.value.code == -1
    buf = e.empty((4, 256))
    e.fft(0, buf, None, None, None, None, None, None, None)  # empty input: no-op
    # ragged trace counts around the wave/block granularity
    for npix in (1, 3, 4, 5, 257):
        x = synth.make_traces(np.arange(npix), 256)
        d = e.to_device(x); f = e.empty((npix, 129, 2))
        e.fft(npix, d, None, None, None, f, None, None, None)
        X = f.download((npix, 129, 2), np.float32)
        ref = np.fft.rfft(x.astype(np.float64), axis=1)
        assert np.abs((X[..., 0] + 1j * X[..., 1]) - ref).max() / np.abs(ref).max() < TOL
        d.free(); f.free()
    buf.free()
    # largest supported power of two
    nt = 16384
    e.set_time_axis(synth.make_time(nt))
    x = synth.make_traces(np.arange(3), nt)
    d = e.to_device(x); f = e.empty((3, nt // 2 + 1, 2)); o = e.empty((3, nt))
    e.fft(3, d, None, None, None, f, None, None, None)
    e.ifft(3, f, None, o, None)
    assert rel(o.download((3, nt), np.float32), x) < TOL
    for b in (d, f, o):
        b.free()


def test_stage_timing_reports_device_time(engine):
    engine.set_time_axis(synth.make_time(1024))
    engine.enable_timing(True)
    x = synth.make_traces(np.arange(64), 1024)
    d = engine.to_device(x); f = engine.empty((64, 513, 2))
    engine.fft(64, d, None, None, None, f, None, None, None)
    assert engine.stage_time_ns(pkg.binding.STAGE_FFT) > 0
    engine.enable_timing(False)
    d.free(); f.free()


# ---- tilt compensation (K11): changes Nt, re-plans ---------------------------
@pytest.mark.parametrize("tilt", [(10.0, 0.0), (0.0, 0.0), (-6.0, 4.0)])
def test_tilt_compensation_then_replan(engine, tilt):
    """tilt_compensation.rs:97-226: extended axis, per-pixel integer shift (bit-exact),
    then the chain continues on the new (non power-of-two) length like
    data_thread.rs:1194-1227"""
    nx, ny, nt = 6, 5, 1024
    time, cube = synth.make_cube(nx, ny, nt)
    dx = dy = 0.5
    steps, new_time, ins = pkg.host_tilt_plan(time, nx, ny, tilt[0], tilt[1], dx, dy)
    osteps, otime, oext = ob.tilt(cube, time, tilt[0], tilt[1], dx, dy)
    assert steps == osteps and np.array_equal(new_time, otime)
    nt2 = nt + 2 * steps
    taper = pkg.host_adapted_blackman(time, 0.0, 7.0)
    e = engine
    d_in = e.to_device(cube); d_tp = e.to_device(taper); d_ins = e.to_device(ins); d_out = e.empty((nx * ny, nt2))
    e.tilt_apply(nx * ny, d_in, nt, d_tp, d_ins, nt2, d_out)
    ext = d_out.download((nx, ny, nt2), np.float32)
    assert np.array_equal(ext, oext)
    # re-plan on the extended axis and run the fft stage there
    e.set_time_axis(new_time)
    assert e.nt == nt2
    assert np.array_equal(e.frequency(), ob.frequency_axis(new_time))
    w = pkg.host_fft_window(new_time, 0, 1.0, 7.0)
    got = gpu_fft_stage(e, ext, w, want_data=False)
    ref = ob.fft_stage(oext, new_time, 0, 1.0, 7.0)
    assert rel(got["fft"], ref["fft"], np.abs(ref["fft"]).max()) < TOL
    for b in (d_in, d_tp, d_ins, d_out):
        b.free()


def test_reference_tilt_unit_tests(engine):
    """tilt_compensation.rs:303-389 through the C ABI"""
    n, impulse_idx = 64, 10
    data = np.zeros((2, 2, n), np.float32)
    data[1, 1, impulse_idx] = 1.0
    time = np.linspace(0.0, np.float32(0.05) * (n - 1), n, dtype=np.float32)
    for tilt_x, expect_ext in ((10.0, True), (0.0, False)):
        steps, new_time, ins = pkg.host_tilt_plan(time, 2, 2, tilt_x, 0.0, 1.0, 1.0)
        assert (steps > 0) == expect_ext
        assert new_time.size == n + 2 * steps
        taper = pkg.host_adapted_blackman(time, 0.0, 7.0)
        d_in = engine.to_device(data); d_tp = engine.to_device(taper); d_ins = engine.to_device(ins)
        d_out = engine.empty((4, n + 2 * steps))
        engine.tilt_apply(4, d_in, n, d_tp, d_ins, n + 2 * steps, d_out)
        out = d_out.download((2, 2, n + 2 * steps), np.float32)
        assert int(np.argmax(out[1, 1])) == impulse_idx + steps
        for b in (d_in, d_tp, d_ins, d_out):
            b.free()


# ---- K13 / K14: build-defined complex / real frequency multipliers ------------
def test_wiener_and_water_lines_chain(engine):
    """reference-pulse Wiener deconvolution + water-line notch as Frequency-domain
    plugins: spectrum * H, then C2R — vs numpy fp64 with the same definitions"""
    nx, ny, nt = 4, 8, 1024
    time, cube = synth.make_cube(nx, ny, nt)
    e = engine
    e.set_time_axis(time)
    freq = e.frequency()
    nf = e.nf
    npix = nx * ny
    # synthetic reference = noise-free pulse template
    z = ((time - time[0] - 11.0) / 0.35).astype(np.float32)
    ref = (-z * np.exp(-z * z)).astype(np.float32)
    w = pkg.host_fft_window(time, 0, 1.0, 7.0)
    d_ref = e.to_device(ref); d_w = e.to_device(w); d_rf = e.empty((nf, 2))
    e.fft(1, d_ref, d_w, None, None, d_rf, None, None, None)
    R = d_rf.download((nf, 2), np.float32)
    H = pkg.host_wiener_filter(R, 1e-2)
    lines = np.loadtxt(os.path.join(GOLD, "water_lines.csv"), dtype=np.float32)
    notch = pkg.host_water_line_mask(freq, lines, 0.01)
    band = pkg.host_fd_bandpass(freq, 0.2, 5.0, 0.1)[0]
    d_x = e.to_device(cube); d_fft = e.empty((npix, nf, 2)); d_amp = e.empty((npix, nf)); d_ph = e.empty((npix, nf))
    e.fft(npix, d_x, d_w, None, None, d_fft, d_amp, d_ph, None)
    amp0 = d_amp.download((npix, nf), np.float32)
    d_H = e.to_device(H); d_n = e.to_device((notch * band).astype(np.float32))
    e.apply_fd_cmask(npix, d_fft, d_amp, d_H)
    e.apply_fd_mask(npix, d_fft, d_amp, d_n)
    d_out = e.empty((npix, nt))
    e.ifft(npix, d_fft, None, d_out, None)
    got = d_out.download((npix, nt), np.float32)
    X = np.fft.rfft(cube.reshape(npix, nt).astype(np.float64) * w, axis=1)
    Hc = (H[:, 0].astype(np.float64) + 1j * H[:, 1]) * (notch * band)
    Y = X * Hc
    Y[:, 0] = Y[:, 0].real
    Y[:, -1] = Y[:, -1].real
    ref_t = np.fft.irfft(Y, n=nt, axis=1)
    assert np.abs(got - ref_t).max() / np.abs(ref_t).max() < 1e-4
    amp = d_amp.download((npix, nf), np.float32)
    assert np.abs(amp - amp0 * np.abs(Hc)).max() / (amp0 * np.abs(Hc)).max() < 1e-4
    F = d_fft.download((npix, nf, 2), np.float32)
    assert np.all(F[:, 0, 1] == 0) and np.all(F[:, -1, 1] == 0)  # C2R precondition (SURVEY a'-4)
    for b in (d_ref, d_w, d_rf, d_x, d_fft, d_amp, d_ph, d_H, d_n, d_out):
        b.free()


def _wiener_from_template(time, eps=1e-2):
    """K13 multiplier of BASELINE config 5, built without the product: the synthetic reference pulse (noise-free
    template) through the fft window, H = conj(R) / (|R|^2 + eps max|R|^2), numpy fp64 -> f32"""
    z = ((time - time[0] - 11.0) / 0.35).astype(np.float64)
    ref = -z * np.exp(-z * z)
    w = ob.apply_window(0, np.ones(time.size, np.float32), time, 1.0, 7.0).astype(np.float64)
    R = np.fft.rfft(ref * w)
    H = np.conj(R) / (np.abs(R) ** 2 + eps * (np.abs(R) ** 2).max())
    out = np.empty((R.size, 2), np.float32)
    out[:, 0] = H.real
    out[:, 1] = H.imag
    return out


@pytest.mark.parametrize("bar", [0, 3])
@pytest.mark.parametrize("mode", ["sums", "cmask", "cmask+sums", "sums+passes", "cmask+sums+passes"])
@pytest.mark.parametrize("shape", [(8, 16, 1024), (3, 7, 2048), (5, 9, 4096), (4, 4, 1001), (3, 5, 1001), (3, 5, 1000), (3, 5, 2000),
                                   (5, 5, 256)])
def test_fused_pipeline_ex(engine, shape, mode, bar, monkeypatch):
    """thz_pipeline_ex: complex per-bin multiplier (K13) inside the fused launch, in-launch pixel sums, store-phase
    barriers — vs the oracle (plain chain) / a numpy fp64 model of the definition in DESIGN.md §7 (K13 is
    build-defined: the reference has no such filter).  Lengths without a fused kernel take the staged fallback."""
    nx, ny, nt = shape
    monkeypatch.setenv("THZ_F_BAR", str(bar))
    if "passes" in mode:   # the F kernels take the sums inside the launch (FSums); this forces the two passes behind it
        monkeypatch.setenv("THZ_NO_FUSED_SUMS", "1")
    time, cube = synth.make_cube(nx, ny, nt)
    e = engine
    e.set_time_axis(time)
    chain_p, chain = synth.default_chain(time), synth.oracle_chain(time)
    npix, nf = nx * ny, nt // 2 + 1
    H = _wiener_from_template(time) if "cmask" in mode else None
    d_raw = e.to_device(cube); d_pre = e.to_device(chain_p["w_pre"]); d_fd = e.to_device(chain_p["fd_mask"])
    d_post = e.to_device(chain_p["w_post"]); d_H = e.to_device(H) if H is not None else None
    d_fft = e.empty((npix, nf, 2)); d_amp = e.empty((npix, nf)); d_ph = e.empty((npix, nf))
    d_out = e.empty((npix, nt)); d_img = e.empty((npix,))
    d_sums = e.empty((2 * nf,)) if "sums" in mode else None
    e.pipeline_ex(npix, d_raw, d_pre, d_fd, d_H, d_post, d_fft, d_amp, d_ph, d_out, d_img, d_sums)
    fft = d_fft.download((npix, nf, 2), np.float32); amp = d_amp.download((npix, nf), np.float32)
    ph = d_ph.download((npix, nf), np.float32); out = d_out.download((npix, nt), np.float32)
    img = d_img.download((npix,), np.float32)
    ref = ob.run_pipeline(cube, time, chain)
    scale = np.abs(ref["fft"]).max()
    if H is None:
        assert rel(fft, ref["fft"].reshape(npix, nf, 2), scale) < TOL
        assert rel(amp, ref["amplitudes"].reshape(npix, nf), scale) < TOL
        assert rel(out, ref["data"].reshape(npix, nt)) < TOL
        assert rel(img, ref["img"].ravel()) < TOL
    else:
        pre = chain["w_tilt"].astype(np.float64) * chain["w_td_before"] * chain["w_fft"]
        X = np.fft.rfft(cube.reshape(npix, nt).astype(np.float64) * pre, axis=1)
        Hc = (H[:, 0].astype(np.float64) + 1j * H[:, 1]) * chain["fd_mask"]
        Y = X * Hc
        a_ref = np.abs(Y)
        Y[:, 0] = Y[:, 0].real
        if nt % 2 == 0:
            Y[:, -1] = Y[:, -1].real
        t_ref = np.fft.irfft(Y, n=nt, axis=1) * chain["w_post"]
        got = fft[..., 0] + 1j * fft[..., 1]
        assert np.abs(got - Y).max() / np.abs(Y).max() < TOL
        assert np.all(fft[:, 0, 1] == 0) and (nt % 2 == 1 or np.all(fft[:, -1, 1] == 0))  # C2R precondition
        assert np.abs(amp - a_ref).max() / a_ref.max() < TOL
        assert np.abs(out - t_ref).max() / np.abs(t_ref).max() < TOL
        assert np.abs(img - (t_ref ** 2).sum(1)).max() / (t_ref ** 2).sum(1).max() < TOL
    st = ob.fft_stage(cube * chain["w_tilt"] * chain["w_td_before"], time, 0, 1.0, 7.0)
    assert phase_ok(ph.reshape(nx, ny, nf), ref["phases"], st["amplitudes"])   # phases are those of X in every mode
    if d_sums is not None:
        sums = d_sums.download((2 * nf,), np.float32)
        sa, sp = amp.astype(np.float64).sum(0), ph.astype(np.float64).sum(0)
        assert np.abs(sums[:nf] - sa).max() <= 2e-6 * np.abs(sa).max()   # f32 sums in another order than the reference's
        assert np.abs(sums[nf:] - sp).max() <= 2e-6 * np.abs(sp).max()
    for b in (d_raw, d_pre, d_fd, d_post, d_H, d_fft, d_amp, d_ph, d_out, d_img, d_sums):
        if b is not None:
            b.free()


# ---- reference pulse ingestion (ConfigCommand::OpenRef, data_thread.rs:372-588) -------------
@pytest.mark.parametrize("nt,shift,nref,wtype", [(1024, 0, 1024, 0), (1024, 40, 1024, 0), (1024, -25, 900, 0),
                                                 (1001, 3, 1200, 0), (2048, 0, 2048, 3), (256, 0, 256, 4)])
def test_reference_spectrum_vs_oracle(engine, nt, shift, nref, wtype):
    scan_t = synth.make_time(nt)
    ref_t = (np.float32(1000.0 + 0.05 * shift) + np.float32(0.05) * np.arange(nref, dtype=np.float32)).astype(np.float32)
    z = (ref_t - (ref_t[0] + np.float32(9.0))) / np.float32(0.35)
    ref_s = (-z * np.exp(-z * z) + 0.002 * np.sin(ref_t)).astype(np.float32)
    got = engine.reference_spectrum(scan_t, ref_t, ref_s, wtype, 1.0, 7.0)
    want = ob.open_ref(scan_t, ref_t, ref_s, wtype, 1.0, 7.0)
    assert want is not None
    assert np.array_equal(got[0], want[0])                       # aligned + windowed pulse: f32 products
    assert rel(got[1], want[1]) < TOL
    # unwrapped phase: bins at the fp32 rounding floor (this pulse has no noise floor of its own)
    # carry no phase information; compare where the amplitude is above 1e-3 of the maximum
    strong = want[1] > 1e-3 * want[1].max()
    d = got[2].astype(np.float64) - want[2]
    jumps = np.round(d / (2 * np.pi))
    assert np.abs(d - 2 * np.pi * jumps)[strong].max() < 3e-3
    lead = np.argmin(strong[1:]) + 1 if not strong[1:].all() else strong.size
    assert not jumps[:lead].any()


def test_reference_spectrum_rejects_what_the_reference_panics_on(engine):
    scan_t = synth.make_time(512)
    ref_t = synth.make_time(400)
    ref_s = np.ones(400, np.float32)
    assert ob.open_ref(scan_t, ref_t, ref_s, 3) is None             # Hamming + unequal lengths: ndarray Zip panic
    with pytest.raises(pkg.ThzError):
        engine.reference_spectrum(scan_t, ref_t, ref_s, 3)
    engine.reference_spectrum(scan_t, ref_t, ref_s, 0)              # the adapted Blackman zips to the shorter one


# ---- chirp-z fused chain (FB kernels) for lengths that are not a power of two -----------------
@pytest.mark.parametrize("shape", [(3, 3, 1000), (5, 1, 513), (2, 2, 300), (7, 1, 77), (1, 1, 1023), (4, 5, 129),
                                   (3, 1, 2000), (2, 2, 1500), (1, 1, 2047), (5, 1, 1025),
                                   (3, 1, 4000), (2, 1, 3000), (1, 1, 4095), (3, 1, 2049),
                                   (3, 1, 5000), (1, 1, 8191), (2, 1, 4097)])
def test_chirpz_fused_pipeline_lengths(engine, shape):
    nx, ny, nt = shape
    time = synth.make_time(nt)
    cube = synth.make_traces(np.arange(nx * ny) + 5, max(nt, 320))[:, :nt].reshape(nx, ny, nt).copy()
    engine.set_kernel_family(2)   # without the mixed-radix kernels: 1000 has one (test_mixed_radix_fused_pipeline)
    try:
        engine.set_time_axis(time)
        assert engine.kernel_variant().startswith("fb-bluestein" if nt < 1024 else ("fb2-" if nt < 2048 else "fb4-" if nt < 4096 else "fb8-"))
        got = synth.run_gpu_pipeline(engine, cube, synth.default_chain(time))
    finally:
        engine.set_kernel_family(0)
    chain = synth.oracle_chain(time)
    ref = ob.run_pipeline(cube, time, chain)
    scale = np.abs(ref["fft"]).max()
    assert rel(got["fft"], ref["fft"], scale) < TOL
    assert rel(got["amplitudes"], ref["amplitudes"], scale) < TOL
    assert rel(got["data"], ref["data"]) < TOL
    assert rel(got["img"], ref["img"]) < TOL
    st = ob.fft_stage(cube * chain["w_tilt"] * chain["w_td_before"], time, 0, 1.0, 7.0)
    assert phase_ok(got["phases"], ref["phases"], st["amplitudes"])


@pytest.mark.parametrize("family", [0, 2])
@pytest.mark.parametrize("shape", [(4, 4, 1001), (5, 1, 1001), (37, 19, 1001), (3, 3, 1000), (64, 33, 1000),
                                   (5, 3, 1200), (7, 3, 1500), (33, 5, 2000), (1, 1, 2000)])
def test_mixed_radix_fused_pipeline(engine, shape, family):
    """nt = 1001 = 7 x 11 x 13, 1000 = 10 x 10 x 10 and the round lengths 1200 / 1500 / 2000 = 10 x 10 x 12 / 15 / 20:
    the P kernels (one direct mixed-radix transform per pair of traces) and, with family 2, the chirp-z kernels they
    replace — same oracle, same tolerances; trace counts odd and even, fewer and more pairs than a block has waves"""
    nx, ny, nt = shape
    time = synth.make_time(nt)
    cube = synth.make_traces(np.arange(nx * ny) + 5, max(nt, 1024))[:, :nt].reshape(nx, ny, nt).copy()
    engine.set_kernel_family(family)
    try:
        engine.set_time_axis(time)
        assert engine.kernel_variant().startswith("p-mixed-radix" if family == 0 else ("fb-bluestein" if nt < 1024 else "fb2-"))
        got = synth.run_gpu_pipeline(engine, cube, synth.default_chain(time))
        # the stage entry points use the same kernels in forward-only / inverse-only form
        st_g = gpu_fft_stage(engine, cube, pkg.host_fft_window(time, 0, 1.0, 7.0))
        d_f = engine.to_device(st_g["fft"]); d_o = engine.empty((nx * ny, nt)); d_i = engine.empty((nx * ny,))
        engine.ifft(nx * ny, d_f, None, d_o, d_i)
        back = d_o.download((nx, ny, nt), np.float32)
        for b in (d_f, d_o, d_i):
            b.free()
    finally:
        engine.set_kernel_family(0)
    chain = synth.oracle_chain(time)
    ref = ob.run_pipeline(cube, time, chain)
    scale = np.abs(ref["fft"]).max()
    assert rel(got["fft"], ref["fft"], scale) < TOL
    assert rel(got["amplitudes"], ref["amplitudes"], scale) < TOL
    assert rel(got["data"], ref["data"]) < TOL
    assert rel(got["img"], ref["img"]) < TOL
    st = ob.fft_stage(cube * chain["w_tilt"] * chain["w_td_before"], time, 0, 1.0, 7.0)
    assert phase_ok(got["phases"], ref["phases"], st["amplitudes"])
    st_o = ob.fft_stage(cube, time, 0, 1.0, 7.0)
    assert np.array_equal(st_g["data"], st_o["data"])
    assert rel(st_g["fft"], st_o["fft"], np.abs(st_o["fft"]).max()) < TOL
    assert phase_ok(st_g["phases"], st_o["phases"], st_o["amplitudes"])
    assert rel(back, st_o["data"]) < TOL   # C2R(R2C(w x)) / nt = w x


@pytest.mark.parametrize("shape", [(3, 3, 2002), (5, 1, 2400), (37, 9, 3000), (33, 5, 4000), (1, 1, 4000), (2, 1, 3000)])
def test_half_length_fused_pipeline(engine, shape):
    """even lengths whose half is a P plan — 2002, 2400, 3000, 4000 — run as a half-length mixed-radix transform +
    split (PH kernels, fft_ph.hpp: one trace per wave, one launch) instead of the chirp-z kernels; same oracle and
    tolerances as every other length, stage entry points included, and the fused launch inverts exactly the spectrum
    it stored"""
    nx, ny, nt = shape
    time = synth.make_time(nt)
    cube = synth.make_traces(np.arange(nx * ny) + 5, max(nt, 1024))[:, :nt].reshape(nx, ny, nt).copy()
    engine.set_time_axis(time)
    assert engine.kernel_variant().startswith("ph-half-length-mixed-radix")
    got = synth.run_gpu_pipeline(engine, cube, synth.default_chain(time))
    st_g = gpu_fft_stage(engine, cube, pkg.host_fft_window(time, 0, 1.0, 7.0))
    d_f = engine.to_device(st_g["fft"]); d_o = engine.empty((nx * ny, nt)); d_i = engine.empty((nx * ny,))
    engine.ifft(nx * ny, d_f, None, d_o, d_i)
    back = d_o.download((nx, ny, nt), np.float32)
    # Filter(6 / 7): the stand-alone inverse on the stored (masked) spectrum lands on the fused launch's samples
    d_f.upload(got["fft"])
    d_w = engine.to_device(synth.default_chain(time)["w_post"])
    engine.ifft(nx * ny, d_f, d_w, d_o, d_i)
    again = d_o.download((nx, ny, nt), np.float32)
    for b in (d_f, d_o, d_i, d_w):
        b.free()
    chain = synth.oracle_chain(time)
    ref = ob.run_pipeline(cube, time, chain)
    scale = np.abs(ref["fft"]).max()
    assert rel(got["fft"], ref["fft"], scale) < TOL
    assert rel(got["amplitudes"], ref["amplitudes"], scale) < TOL
    assert rel(got["data"], ref["data"]) < TOL
    assert rel(got["img"], ref["img"]) < TOL
    st = ob.fft_stage(cube * chain["w_tilt"] * chain["w_td_before"], time, 0, 1.0, 7.0)
    assert phase_ok(got["phases"], ref["phases"], st["amplitudes"])
    st_o = ob.fft_stage(cube, time, 0, 1.0, 7.0)
    assert np.array_equal(st_g["data"], st_o["data"])
    assert rel(st_g["fft"], st_o["fft"], np.abs(st_o["fft"]).max()) < TOL
    assert phase_ok(st_g["phases"], st_o["phases"], st_o["amplitudes"])
    assert rel(back, st_o["data"]) < TOL   # C2R(R2C(w x)) / nt = w x
    assert np.array_equal(again, got["data"])


@pytest.mark.parametrize("shape", [(3, 1, 8193), (2, 3, 10000), (5, 1, 32768), (1, 1, 16386), (2100, 1, 8200)])
def test_long_traces_global_scratch(engine, shape):
    """trace lengths whose transform buffers do not fit the CU's LDS (not a power of two above 8191, powers of two above
    16384; realfft plans any length, io.rs:616-618) run the G kernels with their buffers in global scratch: the fused
    entry point (two launches) and the stage entry points against the oracle; more traces than the scratch has waves"""
    nx, ny, nt = shape
    time = synth.make_time(nt)
    cube = synth.make_traces(np.arange(nx * ny) % 7 + 5, max(nt, 1024))[:, :nt].reshape(nx, ny, nt).copy()
    engine.set_time_axis(time)
    assert "global-scratch" in engine.kernel_variant()
    got = synth.run_gpu_pipeline(engine, cube, synth.default_chain(time))
    chain = synth.oracle_chain(time)
    sel = slice(0, min(nx, 3))          # the oracle's generic transform is slow at these lengths: a few traces
    ref = ob.run_pipeline(cube[sel], time, chain)
    scale = np.abs(ref["fft"]).max()
    assert rel(got["fft"][sel], ref["fft"], scale) < TOL
    assert rel(got["amplitudes"][sel], ref["amplitudes"], scale) < TOL
    assert rel(got["data"][sel], ref["data"]) < TOL
    assert rel(got["img"][sel], ref["img"]) < TOL
    st = ob.fft_stage(cube[sel] * chain["w_tilt"] * chain["w_td_before"], time, 0, 1.0, 7.0)
    assert phase_ok(got["phases"][sel], ref["phases"], st["amplitudes"])
    if nx >= 21:   # every slot of the scratch is reused: the traces repeat with period 7
        assert np.array_equal(got["data"][7:14], got["data"][0:7]) and np.array_equal(got["fft"][-7:], got["fft"][nx - 14:nx - 7])
    st_g = gpu_fft_stage(engine, cube[sel], pkg.host_fft_window(time, 0, 1.0, 7.0))
    st_o = ob.fft_stage(cube[sel], time, 0, 1.0, 7.0)
    assert np.array_equal(st_g["data"], st_o["data"])
    assert rel(st_g["fft"], st_o["fft"], np.abs(st_o["fft"]).max()) < TOL
    assert phase_ok(st_g["phases"], st_o["phases"], st_o["amplitudes"])


@pytest.mark.parametrize("nt", [3, 6, 7, 1001, 2000, 4000])
def test_chirpz_without_windows_matches_numpy(engine, nt):
    """no multipliers at all: X = rfft(x) and y = irfft(X) = x, against numpy fp64"""
    nx, ny = 3, 2
    time = synth.make_time(nt)
    rng = np.random.default_rng(nt)
    cube = (synth.make_traces(np.arange(nx * ny) + 2, max(nt, 320))[:, :nt]
            + 0.1 * rng.standard_normal((nx * ny, nt))).astype(np.float32).reshape(nx, ny, nt)
    engine.set_time_axis(time)
    npix, nf = nx * ny, nt // 2 + 1
    d_raw = engine.to_device(cube)
    d_fft = engine.empty((npix, nf, 2)); d_amp = engine.empty((npix, nf)); d_ph = engine.empty((npix, nf))
    d_out = engine.empty((npix, nt)); d_img = engine.empty((npix,))
    engine.pipeline(npix, d_raw, None, None, None, d_fft, d_amp, d_ph, d_out, d_img)
    X = np.fft.rfft(cube.astype(np.float64), axis=-1)
    G = d_fft.download((nx, ny, nf, 2), np.float32)
    assert np.abs((G[..., 0] + 1j * G[..., 1]) - X).max() / np.abs(X).max() < TOL
    assert rel(d_amp.download((nx, ny, nf), np.float32), np.abs(X)) < TOL
    assert rel(d_out.download((nx, ny, nt), np.float32), cube) < TOL
    assert rel(d_img.download((nx, ny), np.float32), (cube.astype(np.float64) ** 2).sum(-1)) < TOL
    for b in (d_raw, d_fft, d_amp, d_ph, d_out, d_img):
        b.free()


@pytest.mark.parametrize("nt", [256, 1001, 4096])
def test_polar_ifft_of_averaged_spectra(engine, nt):
    """ifft's avg_in_fourier_space branch (math_tools.rs:442-470) and its ROI twin (:496-529): the pixel
    means of amplitudes and unwrapped phases of a small cube, back to one time trace"""
    nx, ny = 4, 3
    time, cube = synth.make_cube(nx, ny, nt)
    st = ob.fft_stage(cube, time, 0, 1.0, 7.0)
    amp, ph = ob.pixel_mean(st["amplitudes"]), ob.pixel_mean(st["phases"])
    engine.set_time_axis(time)
    for zero_dc in (False, True):
        got = engine.polar_ifft(amp, ph, zero_dc)
        ref = ob.polar_irfft(amp, ph, nt, zero_dc)
        assert rel(got, ref) < TOL
    assert np.abs(ref).max() > 0


@pytest.mark.parametrize("cmask", [False, True])
@pytest.mark.parametrize("nt", [1001, 1000, 1024, 2048, 4096, 1500, 2000, 1502, 640])
def test_fused_chain_inverts_exactly_the_stored_spectrum(engine, nt, cmask):
    """The reference's ifft stage reads the fft stage's stored output (data_thread.rs:1090-1105); a Filter(6 / 7)
    update re-runs the stand-alone inverse on the resident spectrum.  So the fused launch's time-domain output must
    be BIT-identical to thz_ifft of the spectrum it stored — no fused multiply-add may swallow the rounding of
    X * mask on the way from the epilogue to the inverse (it did once, in k_p<pipe>: 9e-8 of the maximum)."""
    nx, ny = 5, 7
    time, cube = synth.make_cube(nx, ny, nt)
    e = engine
    e.set_time_axis(time)
    chain = synth.default_chain(time)
    npix, nf = nx * ny, nt // 2 + 1
    H = _wiener_from_template(time) if cmask else None
    d_raw = e.to_device(cube); d_pre = e.to_device(chain["w_pre"]); d_fd = e.to_device(chain["fd_mask"])
    d_post = e.to_device(chain["w_post"]); d_H = e.to_device(H) if cmask else None
    d_fft = e.empty((npix, nf, 2)); d_amp = e.empty((npix, nf)); d_ph = e.empty((npix, nf))
    d_out = e.empty((npix, nt)); d_out2 = e.empty((npix, nt)); d_img = e.empty((npix,)); d_img2 = e.empty((npix,))
    e.pipeline_ex(npix, d_raw, d_pre, d_fd, d_H, d_post, d_fft, d_amp, d_ph, d_out, d_img, None)
    e.ifft(npix, d_fft, d_post, d_out2, d_img2)
    e.sync()
    a, b = d_out.download((npix, nt), np.float32), d_out2.download((npix, nt), np.float32)
    assert np.abs(a).max() > 0
    assert np.array_equal(a, b)
    assert np.array_equal(d_img.download((npix,), np.float32), d_img2.download((npix,), np.float32))


@pytest.mark.parametrize("nt,counts", [(4096, (1, 6, 7, 8, 13, 1792 + 5, 2 * 1792 + 1)), (1024, (1, 9, 2048 + 3, 4096 + 2049)),
                                       (2048, (7, 2048 + 1)), (1001, (1, 2, 3, 33, 8192 + 3, 16384 + 1)), (1000, (5, 8192 + 2)),
                                       (2000, (1, 9, 2048 + 3)), (1500, (7, 6144 + 1))])
def test_in_launch_pixel_sums_ragged_trace_counts(engine, nt, counts):
    """The ticket order of the in-launch pixel sums (FSums / PSums) over trace counts that leave every kind of ragged
    last round: fewer traces than one block has waves, one more than a whole number of rounds of the whole grid, waves
    and whole blocks that never see a trace.  Sums against float64 column sums of the stored arrays; a miscounted
    ticket would hang (bounded: the kernel poisons bin 0 instead) or drop a trace."""
    e = engine
    time = synth.make_time(nt)
    e.set_time_axis(time)
    chain = synth.default_chain(time)
    nf = nt // 2 + 1
    nmax = max(counts)
    d_t = e.to_device(time); d_raw = e.empty((nmax, nt)); e.synth_cube(d_raw, nmax, 0, d_t)
    d_pre = e.to_device(chain["w_pre"]); d_fd = e.to_device(chain["fd_mask"]); d_post = e.to_device(chain["w_post"])
    d_fft = e.empty((nmax, nf, 2)); d_amp = e.empty((nmax, nf)); d_ph = e.empty((nmax, nf)); d_out = e.empty((nmax, nt))
    d_img = e.empty((nmax,)); d_sums = e.empty((2 * nf,))
    for npix in counts:
        e.pipeline_ex(npix, d_raw, d_pre, d_fd, None, d_post, d_fft, d_amp, d_ph, d_out, d_img, d_sums)
        e.sync()
        sums = d_sums.download((2 * nf,), np.float32)
        amp = d_amp.download((nmax, nf), np.float32)[:npix].astype(np.float64)
        ph = d_ph.download((nmax, nf), np.float32)[:npix].astype(np.float64)
        assert np.isfinite(sums).all(), npix
        sa, sp = amp.sum(0), ph.sum(0)
        assert np.abs(sums[:nf] - sa).max() <= 2e-6 * np.abs(sa).max(), npix
        assert np.abs(sums[nf:] - sp).max() <= 2e-6 * max(np.abs(sp).max(), 1.0), npix
    for b in (d_t, d_raw, d_pre, d_fd, d_post, d_fft, d_amp, d_ph, d_out, d_img, d_sums):
        b.free()


@pytest.mark.parametrize("nt,npix", [(4096, 40000), (1001, 50001)])
def test_in_launch_pixel_sums_are_deterministic(engine, nt, npix):
    """The tickets fix the order of every bin's additions (waves of a block in turn, rounds in order, blocks' rows in
    order), so repeated launches give the SAME bits — which an atomics-based accumulation would not.  Twenty launches
    over tens of rounds per block; also a soak of the hand-over itself."""
    e = engine
    time = synth.make_time(nt)
    e.set_time_axis(time)
    chain = synth.default_chain(time)
    nf = nt // 2 + 1
    d_t = e.to_device(time); d_raw = e.empty((npix, nt)); e.synth_cube(d_raw, npix, 0, d_t)
    d_pre = e.to_device(chain["w_pre"]); d_fd = e.to_device(chain["fd_mask"]); d_post = e.to_device(chain["w_post"])
    d_fft = e.empty((npix, nf, 2)); d_amp = e.empty((npix, nf)); d_ph = e.empty((npix, nf)); d_out = e.empty((npix, nt))
    d_img = e.empty((npix,)); d_sums = e.empty((2 * nf,))
    first = None
    for _ in range(20):
        e.pipeline_ex(npix, d_raw, d_pre, d_fd, None, d_post, d_fft, d_amp, d_ph, d_out, d_img, d_sums)
        e.sync()
        sums = d_sums.download((2 * nf,), np.float32)
        assert np.isfinite(sums).all()
        if first is None:
            first = sums
        else:
            assert np.array_equal(sums, first)
    amp = d_amp.download((npix, nf), np.float32).astype(np.float64)
    assert np.abs(first[:nf] - amp.sum(0)).max() <= 2e-6 * np.abs(amp.sum(0)).max()
    for b in (d_t, d_raw, d_pre, d_fd, d_post, d_fft, d_amp, d_ph, d_out, d_img, d_sums):
        b.free()
