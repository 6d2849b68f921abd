"""Host-side pieces of the deconvolution (product C++ through the C ABI, no
GPU) against the oracle's restatement, plus properties of the FIR bank.
tests/golden/psf_sample.npz is the reference's sample_data/psf.npz (data)."""
import os

import numpy as np
import pytest

import oracle_binding as ob
import thz_image_explorer_amd as pkg

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def psfs():
    z = np.load(os.path.join(GOLD, "psf_sample.npz"))
    return pkg.psf_from_npz(z), ob.psf_from_npz(z), z


def test_psf_known_values(psfs):
    """wx_base_a = 0.8575151, wx_base_b = 0.1671958, 20 knots from 0.15 THz"""
    p, o, z = psfs
    assert abs(p.wx_fit.base_a - 0.8575151) < 1e-6 and abs(p.wx_fit.base_b - 0.1671958) < 1e-6
    assert p.wx_fit.correction.n_knots == 20 and abs(float(z["wx_corr_knots_thz"][0]) - 0.15) < 1e-9
    # at a knot the spline returns coeff_a: hybrid = a/f + b + values[k]
    f = np.float32(z["wx_corr_knots_thz"][3])
    wx = pkg.host_psf_eval(p, [f])[0][0]
    expect = np.float32(np.float32(0.8575151) / f + np.float32(0.1671958)) + np.float32(z["wx_corr_values_mm"][3])
    assert abs(wx - expect) < 1e-6


def test_psf_eval_matches_oracle(psfs):
    p, o, _ = psfs
    freqs = np.concatenate([np.geomspace(0.05, 12.0, 200), [0.15, 0.1, 10.0]]).astype(np.float32)
    got = pkg.host_psf_eval(p, freqs)
    ref = ob.psf_eval(o, freqs)
    for g, r in zip(got, ref):
        assert np.array_equal(g, r)
    assert np.all(got[0] > 0) and np.all(got[1] > 0)


@pytest.mark.parametrize("n_filters,f0,f1,width", [(25, 0.1, 10.0, 0.5), (6, 0.4, 3.0, 0.5), (2, 0.5, 2.0, 0.2)])
def test_filter_bank_matches_oracle_and_sums_to_delta(n_filters, f0, f1, width):
    time = (1000 + 0.05 * np.arange(1001)).astype(np.float32)
    cfg = pkg.DeconvCfg(500, n_filters, f0, f1, width)
    filters, centers = pkg.host_filter_bank(time, cfg)
    of, oc = ob.filter_bank(time, n_filters, f0, f1, width)
    assert np.array_equal(centers, oc)
    assert np.array_equal(filters, of)
    # lowpass + bandpasses + highpass telescope to a unit impulse at the centre tap
    s = filters.astype(np.float64).sum(0)
    assert abs(s[249] - 1.0) < 1e-6 and np.abs(np.delete(s, 249)).max() < 1e-6
    assert np.all(np.diff(centers) > 0) and abs(centers[0] - f0) < 1e-6 and abs(centers[-1] - f1) < 1e-5


@pytest.mark.parametrize("dx,dy,rows,cols", [(0.5, 0.5, 32, 32), (1.0, 1.0, 48, 48), (0.25, 0.5, 64, 40)])
def test_band_psf_matches_oracle(psfs, dx, dy, rows, cols):
    p, o, _ = psfs
    for f in (0.25, 0.4, 1.0, 3.0, 9.0):
        got = pkg.host_band_psf(p, f, dx, dy, rows, cols)
        ref = ob.band_psf(o, f, dx, dy, rows, cols)
        assert got.shape == ref.shape and got.shape[0] % 2 == 1 and got.shape[1] % 2 == 1
        assert np.array_equal(got, ref)
        assert abs(got.max() - 1.0) < 0.2  # peak-normalised profiles (psf.rs:244-247), not sum-normalised
