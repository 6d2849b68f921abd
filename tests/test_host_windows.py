"""Product host-side multiplier vectors (csrc/host_windows.cpp through the C
ABI, no GPU) against the oracle's restatement and the reference test
properties."""
import numpy as np
import pytest

import oracle_binding as ob
import thz_image_explorer_amd as pkg


def _axes():
    yield (1000 + 0.05 * np.arange(1024)).astype(np.float32)
    yield (1879 + 0.05 * np.arange(1001)).astype(np.float32)
    yield np.linspace(0.0, 1.0, 128, dtype=np.float32)
    yield np.linspace(-3.0, 40.0, 77, dtype=np.float32)


@pytest.mark.parametrize("wtype", range(5))
def test_fft_window_matches_oracle_bitwise(wtype):
    for time in _axes():
        got = pkg.host_fft_window(time, wtype, 1.0, 7.0)
        ref = ob.apply_window(wtype, np.ones(time.size, np.float32), time, 1.0, 7.0)
        assert np.array_equal(got, ref)


def test_window_properties_reference_test():
    """math_tools.rs:757-840 through the product code"""
    size = 128
    time = np.linspace(0.0, 1.0, size, dtype=np.float32)
    for k in range(5):
        w = pkg.host_fft_window(time, k, 0.1, 0.1)
        assert np.abs(w - w[::-1]).max() <= 1e-5
        if k == 3:
            assert abs(w[0] - 0.08) <= 1e-5
        else:
            assert w[0] <= 1e-5 and w[-1] <= 1e-5
    assert abs(pkg.host_fft_window(time, 0, 0.1, 0.1)[size // 2] - 1.0) <= 1e-5
    assert np.all(pkg.host_fft_window(time, 0, 0.0, 0.0) == 1.0)  # NaN -> 1: identity


@pytest.mark.parametrize("low,high,width", [(0.2, 5.0, 0.1), (0.0, 100.0, 0.0), (-1.0, 3.0, 0.5),
                                            (2.0, 2.05, 0.1), (9.0, 1.0, 0.1)])
def test_fd_bandpass_matches_oracle(low, high, width):
    for time in _axes():
        freq = pkg.host_frequency_axis(time)
        assert np.array_equal(freq, ob.frequency_axis(time))
        got, l, u = pkg.host_fd_bandpass(freq, low, high, width)
        ref, lr, ur = ob.fd_bandpass_window(freq, low, high, width)
        assert (l, u) == (lr, ur)
        assert np.array_equal(got, ref)
        assert np.all(got[:l] == 0) and np.all(got[u:] == 0)


@pytest.mark.parametrize("width", [2.0, 0.1, 0.0])
def test_td_bandpass_matches_oracle(width):
    for time in _axes():
        for low, high in ((float(time[0]), float(time[-1])), (float(time[5]), float(time[40])),
                          (-1e9, 1e9), (float(time[30]), float(time[10]))):
            got, lo, hi, l, u = pkg.host_td_bandpass(time, low, high, width)
            ref, lor, hir, lr, ur = ob.td_bandpass_window(time, low, high, width)
            assert (lo, hi, l, u) == (lor, hir, lr, ur)
            assert np.array_equal(got, ref)


def test_default_td_bandpass_zeroes_last_sample():
    time = (1000 + 0.05 * np.arange(4096)).astype(np.float32)
    w, lo, hi, l, u = pkg.host_td_bandpass(time, float(time[0]), float(time[-1]), 2.0)
    assert (l, u) == (0, 4095) and w[-1] == 0.0


@pytest.mark.parametrize("tilt", [(10.0, 0.0), (0.0, 0.0), (-7.5, 3.25), (15.0, -15.0)])
def test_tilt_plan_matches_oracle(tilt):
    """product host geometry (thz_host_tilt_plan) vs the oracle's restatement of
    tilt_compensation.rs:104-175"""
    nx, ny, nt = 9, 6, 200
    time = (1000 + 0.05 * np.arange(nt)).astype(np.float32)
    data = np.zeros((nx, ny, nt), np.float32)
    data[:, :, 50] = 1.0
    steps, new_time, ins = pkg.host_tilt_plan(time, nx, ny, tilt[0], tilt[1], 0.5, 0.25)
    osteps, otime, out = ob.tilt(data, time, tilt[0], tilt[1], 0.5, 0.25)
    assert steps == osteps
    assert np.array_equal(new_time, otime)
    peak = np.argmax(out.reshape(nx * ny, -1), axis=1)
    assert np.array_equal(ins, peak - 50)


# ---- build-defined frequency multipliers (K13 / K14; not in the reference) -----
def test_water_line_mask_definition():
    import os
    lines = np.loadtxt(os.path.join(os.path.dirname(__file__), "golden", "water_lines.csv"), dtype=np.float32)
    assert lines.size == 135
    time = (1000 + 0.05 * np.arange(4096)).astype(np.float32)
    freq = pkg.host_frequency_axis(time)
    m = pkg.host_water_line_mask(freq, lines, 0.01)
    ref = np.prod(1.0 - np.exp(-(((freq[:, None].astype(np.float64) - lines[None, :]) / 0.01) ** 2)), axis=1)
    assert np.abs(m - ref).max() < 1e-5
    k = np.argmin(np.abs(freq - lines[0]))
    assert m[k] < 0.2 and 0.0 <= m.min() and m.max() <= 1.0
    assert m[np.argmin(np.abs(freq - 0.3))] > 0.999  # far from any line: untouched


def test_wiener_filter_definition():
    rng = np.random.default_rng(0)
    R = (rng.standard_normal(513) + 1j * rng.standard_normal(513)) * np.exp(-np.arange(513) / 100.0)
    r = np.stack([R.real, R.imag], -1).astype(np.float32)
    h = pkg.host_wiener_filter(r, 1e-3)
    H = h[:, 0] + 1j * h[:, 1]
    Rf = r[:, 0].astype(np.float64) + 1j * r[:, 1]
    ref = np.conj(Rf) / (np.abs(Rf) ** 2 + 1e-3 * (np.abs(Rf) ** 2).max())
    assert np.abs(H - ref).max() / np.abs(ref).max() < 1e-5
    strong = np.abs(Rf) > 0.5 * np.abs(Rf).max()
    assert np.abs(H[strong] * Rf[strong] - 1).max() < 5e-3  # inverts where the reference has signal


def test_optical_properties_match_oracle_and_known_answers():
    """calculate_optical_properties, math_tools.rs:663-701"""
    rng = np.random.default_rng(3)
    nf = 513
    freq = (np.arange(nf, dtype=np.float32) / np.float32(51.15)).astype(np.float32)
    ra = rng.uniform(0.1, 5.0, nf).astype(np.float32)
    rp = np.cumsum(rng.uniform(-0.5, 0.0, nf)).astype(np.float32)
    sa = (ra * rng.uniform(0.2, 0.9, nf)).astype(np.float32)
    sp = (rp - rng.uniform(0.0, 3.0, nf)).astype(np.float32)
    sa[5] = 0.0                                   # clamped to 1e-12
    got = pkg.host_optical_properties(sa, sp, ra, rp, freq, 0.7)
    ref = ob.optical_properties(sa, sp, ra, rp, freq, 0.7)
    for g, r in zip(got, ref):
        assert np.array_equal(g, r, equal_nan=True)
    # identical sample and reference: n = 1, alpha = -2/d * ln((2^2)/(4*1)) = 0, kappa = 0 (bin 0: 0/0 -> NaN in n)
    n, alpha, kappa = pkg.host_optical_properties(ra, rp, ra, rp, freq, 1.0)
    assert np.isnan(n[0]) and np.all(n[1:] == 1.0)
    assert np.all(alpha[1:] == 0.0) and np.all(kappa[1:] == 0.0)
    # a phase lag of omega*d/c*(n-1) gives back n: n = 1.5, d = 1 mm... (thickness in metres)
    d = np.float32(1e-3)
    lag = (2 * np.pi * freq.astype(np.float64) * 1e12 * d / 2.99792458e8 * 0.5).astype(np.float32)
    n2, _, _ = pkg.host_optical_properties(ra, rp + lag, ra, rp, freq, float(d))
    assert np.abs(n2[1:] - 1.5).max() < 1e-3


def _pulse(t, tc):
    z = (t - np.float32(tc)) / np.float32(0.35)
    return (-z * np.exp(-z * z)).astype(np.float32)


@pytest.mark.parametrize("case", ["same", "later_start", "earlier_start", "shorter", "longer", "single_point"])
def test_align_reference_matches_oracle(case):
    """ConfigCommand::OpenRef alignment, data_thread.rs:405-481"""
    nt = 400
    scan_t = (np.float32(1000.0) + np.float32(0.05) * np.arange(nt, dtype=np.float32)).astype(np.float32)
    if case == "same":
        ref_t = scan_t.copy()
    elif case == "later_start":       # reference axis starts 37 samples after the scan's -> shifted right
        ref_t = (scan_t + np.float32(0.05 * 37)).astype(np.float32)
    elif case == "earlier_start":     # -> shifted left, head dropped
        ref_t = (scan_t - np.float32(0.05 * 21)).astype(np.float32)
    elif case == "shorter":
        ref_t = scan_t[:300].copy()
    elif case == "longer":
        ref_t = (np.float32(998.0) + np.float32(0.05) * np.arange(520, dtype=np.float32)).astype(np.float32)
    else:
        ref_t = scan_t[:1].copy()
    ref_s = _pulse(ref_t, ref_t[0] + 8.0) if ref_t.size > 1 else np.array([2.5], np.float32)
    got, mode = pkg.host_align_reference(scan_t, ref_t, ref_s)
    ref = ob.open_ref(scan_t, ref_t, ref_s, 0, 0.0, 0.0)     # window bounds 0 -> identity taper
    assert ref is not None
    assert mode == ref[3] == {"same": 0, "single_point": 2}.get(case, 1)
    assert np.array_equal(got, ref[0])
    if case == "later_start":
        assert np.array_equal(got[37:], ref_s[:nt - 37]) and not got[:37].any()
    if case == "earlier_start":
        assert np.array_equal(got[:nt - 21], ref_s[21:])
    if case == "shorter":
        assert np.array_equal(got[:300], ref_s) and not got[300:].any()
