"""Oracle vs the committed numpy-fp64 golden vectors (tests/golden/, made by
scripts/make_golden.py without any reference or oracle code)."""
import os

import numpy as np
import pytest

import oracle_binding as ob

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-5  # north_star: within 1e-5 relative on fp32 spectra


@pytest.fixture(scope="module")
def vec():
    return np.load(os.path.join(GOLD, "fft_vectors.npz"))


@pytest.mark.parametrize("nt", [128, 1000, 1001, 1024, 4096])
def test_fft_stage_all_windows(vec, nt):
    time = vec[f"nt{nt}_time"]
    raw = vec[f"nt{nt}_raw"]
    assert np.array_equal(ob.frequency_axis(time), vec[f"nt{nt}_freq"])
    for kind in range(5):
        st = ob.fft_stage(raw.reshape(1, -1, nt), time, kind, 1.0, 7.0)
        # window values cancel (0.42 - 0.5cos + 0.08cos): f32 cos ulps show up as ~1e-7 absolute
        assert np.abs(st["data"][0] - vec[f"nt{nt}_w{kind}_windowed"]).max() <= 1e-6 * np.abs(raw).max() + 1e-9
        X = st["fft"][0, ..., 0] + 1j * st["fft"][0, ..., 1]
        ref = vec[f"nt{nt}_w{kind}_fft"]
        assert np.abs(X - ref).max() / np.abs(ref).max() < TOL


@pytest.mark.parametrize("nt", [128, 1000, 1001, 1024, 4096])
def test_amp_phase_bandpass_inverse(vec, nt):
    time = vec[f"nt{nt}_time"]
    raw = vec[f"nt{nt}_raw"]
    freq = vec[f"nt{nt}_freq"]
    st = ob.fft_stage(raw.reshape(1, -1, nt), time, 0, 1.0, 7.0)
    amp_ref = vec[f"nt{nt}_w0_amp"]
    assert np.abs(st["amplitudes"][0] - amp_ref).max() / amp_ref.max() < TOL
    # unwrapped phase: compare modulo 2*pi-decision flips on noise bins, and
    # exactly where the signal is strong
    ph_ref = vec[f"nt{nt}_w0_phase"]
    d = st["phases"][0] - ph_ref
    jumps = np.round(d / (2 * np.pi))
    assert np.abs(d - 2 * np.pi * jumps).max() < 2e-3
    strong = amp_ref > 0.05 * amp_ref.max(axis=1, keepdims=True)
    first_weak = np.argmin(strong[:, 5:], axis=1) + 5
    for p in range(raw.shape[0]):
        assert np.all(jumps[p, : first_weak[p]] == 0)
    w, lo, up = ob.fd_bandpass_window(freq, 0.2, 5.0, 0.1)
    assert (lo, up) == tuple(vec[f"nt{nt}_fd_idx"])
    assert np.abs(w - vec[f"nt{nt}_fdmask"]).max() < 1e-6
    f, a = ob.fd_bandpass(st["fft"], st["amplitudes"], freq, 0.2, 5.0, 0.1)
    back, nerr = ob.ifft_stage(f, nt)
    assert nerr == 0
    ref = vec[f"nt{nt}_w0_irfft_bp"]
    assert np.abs(back[0] - ref).max() / np.abs(ref).max() < TOL


def test_mixed_radix_vs_direct_dft():
    rng = np.random.default_rng(5)
    for n in (7, 64, 91, 1001):
        x = rng.standard_normal(n)
        assert np.abs(ob.rfft_f64(x) - ob.rdft_direct_f64(x)).max() < 1e-10
        assert np.abs(ob.rfft_f64(x) - np.fft.rfft(x)).max() < 1e-10


def test_unit_signals():
    u = np.load(os.path.join(GOLD, "unit_signals.npz"))
    X = ob.rfft_f32(u["roundtrip_signal"])
    assert np.abs(X - u["roundtrip_fft"]).max() / np.abs(u["roundtrip_fft"]).max() < TOL
    w, lo, hi, lower, upper = ob.td_bandpass_window(u["td_time"], 0.25, 0.55, 0.0)
    assert (lower, upper) == tuple(u["td_idx"])
    assert np.array_equal(w, u["td_mask"])


def test_roi_masks_bit_exact():
    g = np.load(os.path.join(GOLD, "roi_masks.npz"))
    names = sorted({k[: -len("_poly")] for k in g.files if k.endswith("_poly")})
    assert len(names) == 8
    n_panic = 0
    for name in names:
        poly = g[name + "_poly"]
        for (s0, s1) in ((32, 32), (129, 257)):
            for scaling in (1, 2):
                key = f"{name}_{s0}x{s1}_s{scaling}"
                mask, panic = ob.roi_mask(poly, scaling, s0, s1)
                assert np.array_equal(mask, g[key + "_mask"]), key
                assert panic == bool(g[key + "_panic"]), key
                n_panic += panic
    assert n_panic > 0  # the wrapping rule is exercised


def test_knife_edge_traces_default_chain():
    """16 real traces (Nt = 1001) through the default chain vs numpy fp64"""
    import synth

    k = np.load(os.path.join(GOLD, "knife_edge.npz"))
    time, traces = k["time"], k["traces"]
    assert traces.shape == (16, 1001)
    cube = ob.subtract_bias(traces.reshape(4, 4, 1001))
    chain = synth.oracle_chain(time)
    res = ob.run_pipeline(cube, time, chain)
    pre = chain["w_tilt"].astype(np.float64) * chain["w_td_before"] * chain["w_fft"]
    X = np.fft.rfft(cube.astype(np.float64) * pre, axis=-1)
    Xb = X * chain["fd_mask"]
    got = res["fft"][..., 0] + 1j * res["fft"][..., 1]
    assert np.abs(got - Xb).max() / np.abs(X).max() < TOL
    back = np.fft.irfft(Xb, n=1001, axis=-1) * chain["w_post"]
    assert np.abs(res["data"] - back).max() / np.abs(back).max() < TOL
    assert np.abs(res["img"] - (back ** 2).sum(-1)).max() / (back ** 2).sum(-1).max() < 1e-5
