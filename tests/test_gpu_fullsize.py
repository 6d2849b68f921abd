"""BASELINE.json-size runs on the GPU, checked through size-independent
properties (the oracle is too slow for whole cubes) plus an oracle spot check
on a random sample of traces."""
import numpy as np
import pytest

import oracle_binding as ob
import synth
import thz_image_explorer_amd as pkg

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _device_cube(eng, nx, ny, nt):
    time = synth.make_time(nt)
    eng.set_time_axis(time)
    d_t = eng.to_device(time)
    d = eng.empty((nx * ny, nt))
    eng.synth_cube(d, nx * ny, 0, d_t)
    d_t.free()
    return time, d


@pytest.mark.parametrize("shape", [(256, 256, 1024), (64, 1024, 4096)])
def test_fullsize_properties(engine, shape):
    nx, ny, nt = shape
    npix, nf = nx * ny, nt // 2 + 1
    e = engine
    time, d_raw = _device_cube(e, nx, ny, nt)
    chain = synth.default_chain(time)
    d_pre = e.to_device(chain["w_pre"]); d_fd = e.to_device(chain["fd_mask"]); d_post = e.to_device(chain["w_post"])
    d_fft = e.empty((npix, nf, 2)); d_amp = e.empty((npix, nf)); d_ph = e.empty((npix, nf))
    d_out = e.empty((npix, nt)); d_img = e.empty((npix,))
    e.pipeline(npix, d_raw, d_pre, d_fd, d_post, d_fft, d_amp, d_ph, d_out, d_img)

    # (1) fused == stage-by-stage: the spectrum is bit-identical (same epilogue
    # code); the fused kernel feeds its inverse from registers/LDS in another lane
    # layout, so the time trace agrees to rounding, not bitwise
    f2 = e.empty((npix, nf, 2)); a2 = e.empty((npix, nf)); p2 = e.empty((npix, nf)); o2 = e.empty((npix, nt)); i2 = e.empty((npix,))
    e.fft(npix, d_raw, d_pre, None, None, f2, a2, p2, d_fd)
    e.ifft(npix, f2, d_post, o2, i2)
    fft = d_fft.download((npix, nf, 2), np.float32)
    assert np.array_equal(fft, f2.download((npix, nf, 2), np.float32))
    out = d_out.download((npix, nt), np.float32)
    assert np.abs(out - o2.download((npix, nt), np.float32)).max() / np.abs(out).max() < 2e-6
    img = d_img.download((npix,), np.float32)
    assert np.abs(img - i2.download((npix,), np.float32)).max() / img.max() < 2e-6

    # (2) Parseval on the un-masked transform: sum x^2 = (|X0|^2 + 2 sum |Xk|^2 + |XN|^2)/nt
    e.fft(npix, d_raw, d_pre, None, o2, f2, a2, None, None)
    xw = o2.download((npix, nt), np.float32).astype(np.float64)
    A = a2.download((npix, nf), np.float32).astype(np.float64) ** 2
    lhs = (xw ** 2).sum(1)
    rhs = (A[:, 0] + 2 * A[:, 1:-1].sum(1) + A[:, -1]) / nt
    assert np.abs(lhs - rhs).max() / lhs.max() < 1e-5

    # (3) round trip: ifft(fft(x)) = x (windowed), math_tools.rs:843-897 at full size
    e.ifft(npix, f2, None, o2, None)
    back = o2.download((npix, nt), np.float32)
    assert np.abs(back - xw).max() / np.abs(xw).max() < TOL

    # (4) intensity = sum of squares of the stored trace; amplitudes = |fft|
    assert np.abs(img - (out.astype(np.float64) ** 2).sum(1)).max() / img.max() < TOL
    amp = d_amp.download((npix, nf), np.float32)
    mag = np.hypot(fft[..., 0].astype(np.float64), fft[..., 1].astype(np.float64))
    assert np.abs(amp - mag).max() / mag.max() < TOL
    lo, up = np.nonzero(chain["fd_mask"])[0][[0, -1]]
    assert np.all(fft[:, :lo] == 0) and np.all(fft[:, up + 1:] == 0)  # exact zeros outside the band

    # (5) oracle spot check on 48 random traces
    rng = np.random.default_rng(7)
    idx = np.sort(rng.choice(npix, 48, replace=False))
    raw = synth.make_traces(idx, nt)
    ref = ob.run_pipeline(raw.reshape(1, -1, nt), time, synth.oracle_chain(time))   # the checker's own multiplier vectors
    scale = np.abs(ref["fft"]).max()
    assert np.abs(fft[idx] - ref["fft"][0]).max() / scale < TOL
    assert np.abs(out[idx] - ref["data"][0]).max() / np.abs(ref["data"]).max() < TOL
    assert np.abs(img[idx] - ref["img"][0]).max() / ref["img"].max() < TOL
    ph = d_ph.download((npix, nf), np.float32)[idx]
    d = ph - ref["phases"][0]
    assert np.abs(d - 2 * np.pi * np.round(d / (2 * np.pi))).max() < 5e-3

    # (6) pixel means == mean of the downloaded arrays; linear in the input
    d_avg = e.empty((nf * 2,))
    e.pixel_mean(nx, ny, nf, 2, d_fft, d_avg)
    avg = d_avg.download((nf, 2), np.float32)
    ref_avg = fft.astype(np.float64).mean(0)
    assert np.abs(avg - ref_avg).max() / np.abs(ref_avg).max() < 1e-4
    for b in (d_raw, d_pre, d_fd, d_post, d_fft, d_amp, d_ph, d_out, d_img, f2, a2, p2, o2, i2, d_avg):
        b.free()


def test_linearity_fullsize(engine):
    """fft(a*x + b*y) = a*fft(x) + b*fft(y) on a 128x128x4096 tile"""
    nx, ny, nt = 128, 128, 4096
    npix, nf = nx * ny, nt // 2 + 1
    e = engine
    time = synth.make_time(nt)
    e.set_time_axis(time)
    d_t = e.to_device(time)
    dx = e.empty((npix, nt)); dy = e.empty((npix, nt))
    e.synth_cube(dx, npix, 0, d_t)
    e.synth_cube(dy, npix, 10_000_000, d_t)
    x = dx.download((npix, nt), np.float32); y = dy.download((npix, nt), np.float32)
    z = (np.float32(0.75) * x + np.float32(-1.5) * y).astype(np.float32)
    dz = e.to_device(z)
    fx = e.empty((npix, nf, 2)); fy = e.empty((npix, nf, 2)); fz = e.empty((npix, nf, 2))
    e.fft(npix, dx, None, None, None, fx, None, None, None)
    e.fft(npix, dy, None, None, None, fy, None, None, None)
    e.fft(npix, dz, None, None, None, fz, None, None, None)
    X = fx.download((npix, nf, 2), np.float32).astype(np.float64)
    Y = fy.download((npix, nf, 2), np.float32).astype(np.float64)
    Z = fz.download((npix, nf, 2), np.float32).astype(np.float64)
    lin = 0.75 * X - 1.5 * Y
    assert np.abs(Z - lin).max() / np.abs(lin).max() < TOL
    for b in (d_t, dx, dy, dz, fx, fy, fz):
        b.free()
