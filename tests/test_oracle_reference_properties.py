"""The reference's own on-path unit tests, re-run
against the CPU oracle: they are the only pins the reference offers for this
path, so the oracle has to pass them before it may judge the HIP code."""
import numpy as np
import pytest

import oracle_binding as ob

WIN_ADAPTED, WIN_BLACKMAN, WIN_HANN, WIN_HAMM, WIN_FLAT = range(5)


def test_window_functions_apply():
    """math_tools.rs:757-840"""
    size = 128
    time = np.linspace(0.0, 1.0, size, dtype=np.float32)
    ones = np.ones(size, np.float32)
    sig = {k: ob.apply_window(k, ones, time) for k in (WIN_BLACKMAN, WIN_HANN, WIN_HAMM, WIN_FLAT)}
    sig[WIN_ADAPTED] = ob.apply_adapted_blackman(ones, time, 0.1, 0.1)
    for k in (WIN_BLACKMAN, WIN_HANN, WIN_FLAT, WIN_ADAPTED):
        assert sig[k][0] <= 1e-5 and sig[k][-1] <= 1e-5
    assert abs(sig[WIN_HAMM][0] - 0.08) <= 1e-5 and abs(sig[WIN_HAMM][-1] - 0.08) <= 1e-5
    for k, s in sig.items():
        assert np.abs(s - s[::-1]).max() <= 1e-5, k
        mid = size // 2
        assert s[mid] >= s[mid - 1] and s[mid] >= s[mid + 1], k
    assert abs(sig[WIN_ADAPTED][size // 2] - 1.0) <= 1e-5


def test_fft_roundtrip():
    """math_tools.rs:843-897: ifft(fft(x)).data == fft(x).data within 1e-4 abs,
    window [0,0] (identity), unnormalised forward, /N inverse."""
    n = 128
    tt = np.arange(n, dtype=np.float32) / np.float32(n)
    data = (np.sin(2 * np.float32(np.pi) * 3 * tt) + 0.5 * np.cos(2 * np.float32(np.pi) * 7 * tt)).astype(np.float32)
    time = np.linspace(0.0, 1.0, n, dtype=np.float32)
    st = ob.fft_stage(data.reshape(1, 1, n), time, WIN_ADAPTED, 0.0, 0.0)
    assert np.array_equal(st["data"].ravel(), data)  # window [0,0] is the identity (NaN -> 1)
    back, nerr = ob.ifft_stage(st["fft"], n)
    assert nerr == 0
    assert np.abs(back.ravel() - st["data"].ravel()).max() <= 1e-4
    # pins "unnormalised forward": bin 3 of sin has magnitude n/2
    amp = st["amplitudes"].ravel()
    assert abs(amp[3] - n / 2) < 1e-3 and abs(amp[7] - n / 4) < 1e-3


def test_fd_bandpass_exact_zeros():
    """band_pass_fd.rs:475-567: 1x1x256 sine at bin 9, freq i/50, pass +-2 bins, width 0"""
    n, k = 256, 9
    tt = np.arange(n, dtype=np.float32)
    sig = np.sin(2 * np.float32(np.pi) * k * tt / n).astype(np.float32)
    time = np.linspace(0.0, 1.0, n, dtype=np.float32)
    st = ob.fft_stage(sig.reshape(1, 1, n), time, WIN_ADAPTED, 0.0, 0.0)
    freq = (np.arange(n // 2 + 1, dtype=np.float32) / np.float32(50.0)).astype(np.float32)
    low, high = float(freq[k - 2]), float(freq[k + 2])
    f, a = ob.fd_bandpass(st["fft"], st["amplitudes"], freq, low, high, 0.0)
    assert f.shape == st["fft"].shape and a.shape == st["amplitudes"].shape
    _, lower, upper = ob.fd_bandpass_window(freq, low, high, 0.0)
    a = a.ravel()
    assert np.all(a[:lower] == 0.0) and np.all(a[upper:] == 0.0)
    assert a[lower:upper].sum() > 0.0
    assert np.array_equal(st["phases"], st["phases"])  # phases pass through untouched


@pytest.mark.parametrize("width_default", [2.0, 0.1])
def test_td_bandpass_exact_zeros(width_default):
    """band_pass_td_before_fft.rs:390-443 / band_pass_td_after_fft.rs:389-"""
    n = 256
    time = np.linspace(0.0, 1.0, n, dtype=np.float32)
    sig = np.sin(2 * np.float32(np.pi) * 5 * time).astype(np.float32)
    out, lo, hi = ob.td_bandpass(sig.reshape(1, 1, n), time, 0.25, 0.55, 0.0)
    _, _, _, lower, upper = ob.td_bandpass_window(time, 0.25, 0.55, 0.0)
    out = out.ravel()
    assert out.shape == sig.shape
    assert np.all(out[:lower] == 0.0) and np.all(out[upper:] == 0.0)
    assert np.abs(out[lower:upper]).sum() > 0.0
    assert np.array_equal(out[lower:upper], sig[lower:upper])  # width 0: NaN -> 1 inside


def test_default_time_bandpass_is_not_identity():
    """SURVEY a'-1: after reset() low/high are the axis ends, upper = Nt-1, so
    the last sample is zeroed and both ends are tapered."""
    nt = 1024
    time = (1000 + 0.05 * np.arange(nt)).astype(np.float32)
    w, lo, hi, lower, upper = ob.td_bandpass_window(time, float(time[0]), float(time[-1]), 2.0)
    assert (lower, upper) == (0, nt - 1)
    assert w[-1] == 0.0 and w[0] == 0.0 and w[nt // 2] == 1.0


def test_unwrap_recurrence():
    """math_tools.rs:211-240 against the same recurrence in float64"""
    rng = np.random.default_rng(3)
    ph = rng.uniform(-np.pi, np.pi, 513).astype(np.float32)
    got = ob.numpy_unwrap(ph)
    ref = np.empty(ph.size)
    ref[0] = ph[0]
    for i in range(1, ph.size):
        d = float(ph[i]) - float(ph[i - 1])
        if d > np.float32(np.pi): d -= 2 * float(np.float32(np.pi))
        elif d < -np.float32(np.pi): d += 2 * float(np.float32(np.pi))
        ref[i] = ref[i - 1] + d
    assert np.abs(got - ref).max() < 1e-3
    assert np.all(np.abs(np.diff(got)) <= np.pi + 1e-5)


def test_frequency_axis_rule():
    """io.rs:614-621: i / (t_last - t_first), not i / (nt*dt)"""
    time = (1879 + 0.05 * np.arange(1001)).astype(np.float32)
    f = ob.frequency_axis(time)
    assert f.size == 501
    assert np.allclose(f, np.arange(501, dtype=np.float32) / (time[-1] - time[0]), rtol=0, atol=0)


def test_scaling_block_mean_ragged():
    """math_tools.rs:273-301 (a'-6): floor(nx/s), divide by s*s always"""
    rng = np.random.default_rng(0)
    a = rng.standard_normal((5, 7, 6)).astype(np.float32)
    out = ob.scale3d(a, 2)
    assert out.shape == (2, 3, 6)
    ref = a[:4, :6].reshape(2, 2, 3, 2, 6).sum(axis=(1, 3)) / 4
    assert np.abs(out - ref).max() < 1e-6


def test_intensity_and_bias():
    rng = np.random.default_rng(1)
    a = rng.standard_normal((3, 4, 64)).astype(np.float32)
    b = ob.subtract_bias(a)
    assert np.all(b[..., 0] == 0.0)
    img = ob.intensity(b)
    assert np.allclose(img, (b.astype(np.float64) ** 2).sum(-1), rtol=1e-5)


def _expected_tilt_steps(width, height, dx, dy, tx_deg, ty_deg):
    """compute_expected_num_steps of the reference's test module (tilt_compensation.rs:283-300)"""
    f = np.float32
    tsx = f(f(tx_deg) / f(180.0) * f(np.pi)); tsy = f(f(ty_deg) / f(180.0) * f(np.pi))
    cx = f(f(width) / f(2.0) * f(dx)); cy = f(f(height) / f(2.0) * f(dy))
    c = 0.299792458
    mx = f(float(cx) * float(abs(tsx)) / c); my = f(float(cy) * float(abs(tsy)) / c)
    ext = f(np.floor(f(f(mx + my) / f(0.05)))) * f(0.05)
    return int(np.round(f(ext / f(0.05))))


def test_tilt_extends_time_and_shifts_center_trace():
    """tilt_compensation.rs:303-346 (known-answer, integer indices)"""
    n, dt, impulse_idx = 64, np.float32(0.05), 10
    data = np.zeros((2, 2, n), np.float32)
    data[1, 1, impulse_idx] = 1.0
    time = np.linspace(0.0, dt * (n - 1), n, dtype=np.float32)
    steps, new_time, out = ob.tilt(data, time, 10.0, 0.0, 1.0, 1.0)
    exp = _expected_tilt_steps(2, 2, 1.0, 1.0, 10.0, 0.0)
    assert steps == exp and exp > 0
    assert new_time.size == n + 2 * exp and out.shape[-1] == n + 2 * exp
    assert int(np.argmax(out[1, 1])) == impulse_idx + exp  # centre pixel: zero geometric shift
    assert np.all(np.diff(new_time) > 0)


def test_tilt_no_tilt_no_extension():
    """tilt_compensation.rs:349-389"""
    n, dt, impulse_idx = 64, np.float32(0.05), 12
    data = np.zeros((2, 2, n), np.float32)
    data[1, 1, impulse_idx] = 1.0
    time = np.linspace(0.0, dt * (n - 1), n, dtype=np.float32)
    steps, new_time, out = ob.tilt(data, time, 0.0, 0.0, 1.0, 1.0)
    assert steps == 0 and new_time.size == n
    assert int(np.argmax(out[1, 1])) == impulse_idx
