"""BASELINE.json configs 3, 4 and 5 as written, at their workloads, through the session API the data thread
would drive ("configs_untested").  Stand-ins for the files that are Git-LFS pointers in the
reference checkout are the ones SURVEY.md §8d names: a 128 x 128 x 1001 cube tiled from the 16 real
knife-edge traces (tests/golden/knife_edge.npz) times a bar-target mask for `resolution_target_sample.thzimg`,
the noise-free pulse template for `reference.thz`, the reference's own `sample_data/psf.npz`
(tests/golden/psf_sample.npz).

K13 (reference-pulse Wiener filter) and K14 (water-line notch) are build-defined (the reference
has neither), so their checker is a numpy fp64 model of the stated definition on top of the oracle's chain."""
import os

import numpy as np
import pytest

import oracle_binding as ob
import synth
import thz_image_explorer_amd as pkg
from test_gpu_parity import TOL, phase_ok, phase_parity, rel

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def resolution_target_stand_in(nx=128, ny=128):
    """(time, cube): real knife-edge traces (nt = 1001, bias-subtracted like open_scan_from_thz does) tiled over
    the grid with per-pixel amplitudes of a bar target + seeded 1 % jitter"""
    k = np.load(os.path.join(GOLD, "knife_edge.npz"))
    time, traces = k["time"], ob.subtract_bias(k["traces"].reshape(4, 4, 1001)).reshape(16, 1001)
    xx, yy = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    bars = 0.35 + 0.65 * (((xx // 3) % 2 == 0) & (yy > ny // 4) & (yy < 3 * ny // 4))
    bars = bars + 0.3 * (((yy // 5) % 2 == 0) & (xx > nx // 2))
    rng = np.random.default_rng(0x7A3D2026)
    amp = (bars * (1.0 + 0.01 * rng.standard_normal((nx, ny)))).astype(np.float32)
    which = ((xx * 7 + yy * 3) % 16).astype(np.int64)
    cube = (traces[which] * amp[..., None]).astype(np.float32)
    return time, np.ascontiguousarray(cube)


def water_line_notch_model(freq, sigma=0.01):
    """K14 by its definition, in fp64: prod_i (1 - exp(-((f - f_i) / sigma)^2)) over assets/water_lines.csv"""
    lines = np.loadtxt(os.path.join(GOLD, "water_lines.csv"), dtype=np.float64)
    lines = lines.astype(np.float32).astype(np.float64)    # the f32 values the product is handed
    f = freq.astype(np.float64)[:, None]
    return np.prod(1.0 - np.exp(-((f - lines[None, :]) / np.float64(np.float32(sigma))) ** 2), axis=1)


# ---------------------------------------------------------------------------------------- config 3
def test_config3_resolution_target_full_chain_with_water_line_filter(engine):
    """`resolution_target_sample.thzimg` (stand-in) through the FULL session chain — tilt taper, Time Band Pass, fft
    window, R2C, Frequency Band Pass x water-line notch (a FilterDomain::Frequency plugin behind it), C2R, Time
    Band Pass, image, pixel means — on one MI355X, fp32 parity vs the oracle chain with the notch from its
    definition"""
    nx = ny = 128
    time, cube = resolution_target_stand_in(nx, ny)
    nt, nf = time.size, time.size // 2 + 1
    freq = ob.frequency_axis(time)
    lines = np.loadtxt(os.path.join(GOLD, "water_lines.csv"), dtype=np.float32)
    notch = pkg.host_water_line_mask(freq, lines, 0.01)                      # the product's K14
    model = water_line_notch_model(freq)
    assert np.abs(notch - model).max() < 2e-6 and notch.min() < 0.05 and notch.max() > 0.99
    sess = pkg.Session(engine, nx, ny, time, 0.5, 0.5)
    try:
        sess.upload(cube, subtract_bias=False)
        assert engine.kernel_variant().startswith("p-mixed-radix-7x11x13")
        sess.set_fd_filters(notch, None)
        cfg = pkg.chain_cfg_default(time)
        sess.recompute(cfg)
        chain = synth.oracle_chain(time)
        chain["fd_mask"] = (chain["fd_mask"] * model.astype(np.float32)).astype(np.float32)
        ref = ob.run_pipeline(cube, time, chain)
        scale = np.abs(ref["fft"]).max()
        got_fft = sess.download(pkg.BUF_FFT).reshape(nx, ny, nf, 2)
        assert rel(got_fft, ref["fft"], scale) < TOL
        assert rel(sess.download(pkg.BUF_AMPLITUDES).reshape(nx, ny, nf), ref["amplitudes"], scale) < TOL
        assert rel(sess.download(pkg.BUF_DATA).reshape(nx, ny, nt), ref["data"]) < TOL
        assert rel(sess.download(pkg.BUF_IMG).reshape(nx, ny), ref["img"]) < TOL
        st = ob.fft_stage(cube * chain["w_tilt"] * chain["w_td_before"], time, 0, 1.0, 7.0)
        ok, why = phase_parity(sess.download(pkg.BUF_PHASES).reshape(nx, ny, nf), ref["phases"], st["fft"])
        assert ok, why
        # the notch really bites: bins on a strong line are gone, the pass band between lines is not
        k_line = int(np.argmin(np.abs(freq - 1.0974)))      # 1.097 THz water line
        assert np.abs(got_fft[:, :, k_line]).max() < 0.05 * np.abs(got_fft).max()
        # pixel means of the ifft stage (fast form: sums + linearity) against the reference's summation order
        for which, ncomp, key in ((pkg.BUF_AVG_FFT, 2, "fft"), (pkg.BUF_AVG_AMPLITUDES, 1, "amplitudes")):
            want = ob.pixel_mean(ref[key], ncomp)
            assert rel(sess.download(which), want) < TOL
        # mean unwrapped phase: a 2 pi decision flip on a noise bin of ONE trace (tolerated above, and unpinned by
        # the reference itself) moves the mean by 2 pi / 16 384, so the oracle's mean is only good to the flips it
        # took; the device's mean must be the mean of the device's own phases
        got_ph = sess.download(pkg.BUF_PHASES).astype(np.float64)
        assert rel(sess.download(pkg.BUF_AVG_PHASES), got_ph.mean(0)) < 2e-6
        flips = np.abs(np.round((got_ph.reshape(nx, ny, nf) - ref["phases"]) / (2 * np.pi))).mean(axis=(0, 1))
        assert np.all(np.abs(sess.download(pkg.BUF_AVG_PHASES) - ob.pixel_mean(ref["phases"], 1))
                      <= 2 * np.pi * flips + 1e-5 * np.abs(ref["phases"]).max())
    finally:
        sess.close()


# ---------------------------------------------------------------------------------------- config 4
def test_config4_psf_deconvolution_reference_defaults(engine):
    """stand-in + psf.npz, Deconvolution with the reference's defaults — 500 iterations, 25 bands, 0.1-10 THz, 0.5
    window (deconvolution.rs:725-734) — as the chain's last stage, vs the oracle, at the north star's 1e-5.  Error
    budget (scripts/gpu_deconv_error_budget.py): the device's FIR and band energies run in fp32
    where the reference uses a Complex<f64> FFT (deconvolution.rs:266-317); that costs 3e-7 on the cube, and the 423
    Richardson-Lucy iterations of the widest band do not amplify it (1.6e-6 after all 500)."""
    nx = ny = 64
    time, cube = resolution_target_stand_in(nx, ny)
    nt = time.size
    z = np.load(os.path.join(GOLD, "psf_sample.npz"))
    psf, opsf = pkg.psf_from_npz(z), ob.psf_from_npz(z)
    sess = pkg.Session(engine, nx, ny, time, 0.5, 0.5)
    try:
        sess.upload(cube, subtract_bias=False)
        cfg = pkg.chain_cfg_default(time)
        sess.recompute(cfg)
        chain_out = sess.download(pkg.BUF_DATA).reshape(nx, ny, nt).copy()
        dcfg = pkg.DeconvCfg(500, 25, 0.1, 10.0, 0.5)
        assert sess.deconvolve(psf, dcfg) == 0
        out = sess.download(pkg.BUF_DATA).reshape(nx, ny, nt)
        img = sess.download(pkg.BUF_IMG).reshape(nx, ny)
        # the oracle deconvolves the session's own Time Band Pass output: this test is about the last stage
        rc, oref, oimg, ogains, oniter = ob.deconvolution(chain_out, time, 0.5, 0.5, opsf, 500, 25, 0.1, 10.0, 0.5)
        assert rc == 0 and oniter.max() >= 400 and oniter.min() >= 1
        assert np.isfinite(out).all()
        assert rel(out, oref) < 1e-5       # measured 1.6e-6 (profiles/r02_deconv_error_budget.txt)
        assert rel(img, oimg) < 1e-5       # measured 1.5e-6
        assert np.abs(out - chain_out).max() / np.abs(chain_out).max() > 1e-2   # it is not the identity
        # any other filter update passes the stage through again (data_thread.rs:1139-1149)
        sess.recompute(cfg, 7)
        assert np.array_equal(sess.download(pkg.BUF_DATA).reshape(nx, ny, nt), chain_out)
    finally:
        sess.close()


# ---------------------------------------------------------------------------------------- config 5
def _reference_pulse_file(time):
    """what `reference.thz` would hold: the noise-free pulse template on ITS OWN axis — starts 3 ps earlier, has 200
    samples fewer — so that OpenRef's index-shift alignment (data_thread.rs:405-481) has work to do"""
    dt = np.float32(0.05)
    ref_t = (np.float32(time[0] - 3.0) + dt * np.arange(time.size - 200, dtype=np.float32)).astype(np.float32)
    zz = ((ref_t - time[0] - 11.0) / 0.35).astype(np.float32)
    return ref_t, (-zz * np.exp(-zz * zz)).astype(np.float32)


@pytest.mark.parametrize("shape", [(64, 1024, 4096), (16, 16, 1024)])
def test_config5_synthetic_cube_with_reference_wiener_filter(engine, shape):
    """synthetic cube + reference-pulse Wiener deconvolution (K13) in the fused launch: the reference pulse is
    ingested by the OpenRef rules (aligned to the scan axis, windowed with the reference file's own axis,
    transformed with the scan's plan), H = conj(R) / (|R|^2 + eps max|R|^2) rides in thz_pipeline_ex as the complex
    per-bin multiplier.  Size-independent properties + a 48-trace spot check against numpy fp64 on the oracle's
    ingestion of the same pulse."""
    nx, ny, nt = shape
    npix, nf = nx * ny, nt // 2 + 1
    e = engine
    time = synth.make_time(nt)
    ref_t, ref_s = _reference_pulse_file(time)
    # ---- product side: OpenRef through the C ABI, then K13 from the engine's transform of the ingested pulse
    ref_p, amp_p, ph_p = e.reference_spectrum(time, ref_t, ref_s, 0, 1.0, 7.0)
    d_ref = e.to_device(ref_p); d_R = e.empty((nf, 2))
    e.fft(1, d_ref, None, None, None, d_R, None, None, None)
    H = pkg.host_wiener_filter(d_R.download((nf, 2), np.float32), 1e-2)
    d_ref.free(); d_R.free()
    # ---- checker side: the oracle's OpenRef, the filter from its definition in fp64
    oref, oamp, oph, mode = ob.open_ref(time, ref_t, ref_s, 0, 1.0, 7.0)
    assert mode == 1                                     # shifted, not naive
    assert np.array_equal(ref_p, oref)
    Ro = np.fft.rfft(oref.astype(np.float64))
    Ho = np.conj(Ro) / (np.abs(Ro) ** 2 + 1e-2 * (np.abs(Ro) ** 2).max())
    assert np.abs((H[:, 0] + 1j * H[:, 1]) - Ho).max() / np.abs(Ho).max() < 1e-5

    sess = pkg.Session(e, nx, ny, time)
    try:
        d_t = e.to_device(time)
        e.synth_cube(sess.eng.lib.thz_session_buffer(sess.h, pkg.BUF_RAW), npix, 0, d_t)
        d_t.free()
        sess.upload(None, subtract_bias=False)
        sess.set_fd_filters(None, H)
        cfg = pkg.chain_cfg_default(time)
        sess.recompute(cfg)
        fft = sess.download(pkg.BUF_FFT).reshape(npix, nf, 2)
        amp = sess.download(pkg.BUF_AMPLITUDES).reshape(npix, nf)
        out = sess.download(pkg.BUF_DATA).reshape(npix, nt)
        img = sess.download(pkg.BUF_IMG).ravel()
        # (1) C2R precondition; exact zeros outside the band; amplitudes = |stored spectrum| off DC / Nyquist
        assert np.all(fft[:, 0, 1] == 0) and np.all(fft[:, -1, 1] == 0)
        chain = synth.oracle_chain(time)
        lo, up = np.nonzero(chain["fd_mask"])[0][[0, -1]]
        assert np.all(fft[:, :lo] == 0) and np.all(fft[:, up + 1:] == 0)
        mag = np.hypot(fft[..., 0].astype(np.float64), fft[..., 1].astype(np.float64))
        assert np.abs(amp[:, 1:-1] - mag[:, 1:-1]).max() / mag.max() < TOL
        # (2) the stored trace is the inverse transform of the stored spectrum: one more C2R through the stage entry point
        d_o2 = e.empty((npix, nt)); d_i2 = e.empty((npix,))
        d_post = e.to_device(chain["w_post"])
        e.ifft(npix, e.lib.thz_session_buffer(sess.h, pkg.BUF_FFT), d_post, d_o2, d_i2)
        assert np.abs(out - d_o2.download((npix, nt), np.float32)).max() / np.abs(out).max() < 2e-6
        for b in (d_o2, d_i2, d_post):
            b.free()
        assert np.abs(img - (out.astype(np.float64) ** 2).sum(1)).max() / img.max() < TOL
        # (3) 48 random traces against numpy fp64 on the oracle's multipliers
        rng = np.random.default_rng(5)
        idx = np.sort(rng.choice(npix, min(48, npix), replace=False))
        raw = synth.make_traces(idx, nt).astype(np.float64)
        pre = chain["w_tilt"].astype(np.float64) * chain["w_td_before"] * chain["w_fft"]
        Y = np.fft.rfft(raw * pre, axis=1) * (Ho * chain["fd_mask"])
        a_ref = np.abs(Y)
        Y[:, 0] = Y[:, 0].real
        Y[:, -1] = Y[:, -1].real
        t_ref = np.fft.irfft(Y, n=nt, axis=1) * chain["w_post"]
        got = fft[idx, :, 0] + 1j * fft[idx, :, 1]
        assert np.abs(got - Y).max() / np.abs(Y).max() < TOL
        assert np.abs(amp[idx] - a_ref).max() / a_ref.max() < TOL
        assert np.abs(out[idx] - t_ref).max() / np.abs(t_ref).max() < TOL
        # the Wiener filter does what it is for: the filtered main pulse is narrower than the band-passed one
        sess.set_fd_filters(None, None)
        sess.recompute(cfg)
        plain = sess.download(pkg.BUF_DATA, 0, idx[0] + 1)[idx[0]]

        def width(v):
            return int((np.abs(v) > 0.5 * np.abs(v).max()).sum())
        assert width(out[idx[0]]) < width(plain)
        # (4) pixel means with the complex multiplier: avg_fft by linearity == mean of the stored spectra
        sess.set_fd_filters(None, H)
        sess.recompute(cfg)
        avg = sess.download(pkg.BUF_AVG_FFT)
        want = fft.astype(np.float64).mean(0)
        assert np.abs(avg - want).max() / np.abs(want).max() < 1e-5
        # ... and the amplitude / phase means, summed inside the same launch (k_f<pipe, complex multiplier, sums>)
        want_a = amp.astype(np.float64).mean(0)
        assert np.abs(sess.download(pkg.BUF_AVG_AMPLITUDES) - want_a).max() / want_a.max() < 2e-6
        want_p = sess.download(pkg.BUF_PHASES).reshape(npix, nf).astype(np.float64).mean(0)
        assert np.abs(sess.download(pkg.BUF_AVG_PHASES) - want_p).max() / np.abs(want_p).max() < 2e-6
    finally:
        sess.close()
