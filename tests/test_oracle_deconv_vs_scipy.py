"""Independent pins for the oracle's deconvolution building blocks: numpy / scipy float64 models made of
library primitives (firwin, convolve, correlate2d / convolve2d), none of the oracle's or the product's code.
They fix the conventions a restatement can get wrong silently: the Kaiser design and spectral inversion of
create_filter_bank (deconvolution.rs:30-211), the 'same' slice of the FIR output (shift (taps-1)/2), and the two
2-D convolution branches of richardson_lucy — kernels of at most 256 elements are CORRELATION-indexed
(:432-458), larger ones true 'same' convolutions (the FFT branch) — on the reflect-padded image, with
eps = 1e-12 and no renormalisation."""
import numpy as np
import pytest
from scipy import signal

import oracle_binding as ob
import synth


@pytest.mark.parametrize("n_filters,f0,f1,width", [(25, 0.1, 10.0, 0.5), (6, 0.4, 3.0, 0.5), (3, 0.5, 2.0, 0.2)])
def test_filter_bank_against_scipy_firwin(n_filters, f0, f1, width):
    nt = 1001
    time = synth.make_time(nt)
    filters, centers = ob.filter_bank(time, n_filters, f0, f1, width)
    fs = 1.0 / float(time[1] - time[0])
    taps = filters.shape[1]
    cen = np.exp(np.linspace(np.log(f0), np.log(f1), n_filters))
    assert np.abs(centers - cen).max() / cen.max() < 1e-6
    beta = signal.kaiser_beta(signal.kaiser_atten(taps, width / (0.5 * fs)))

    def lowpass(fc):
        # scipy's Kaiser window ends at 1/I0(beta); the reference sets the window's two end points to exactly 0
        # (deconvolution.rs: `if n != 0 && n != adj - 1`) and then normalises the sum to 1
        h = signal.firwin(taps, fc, window=("kaiser", beta), fs=fs, scale=False)
        h[0] = h[-1] = 0.0
        return h / h.sum()

    for i in range(n_filters):
        lo = 0.0 if i == 0 else np.sqrt(cen[i - 1] * cen[i])
        hi = 0.5 * fs if i == n_filters - 1 else np.sqrt(cen[i] * cen[i + 1])
        delta = np.zeros(taps); delta[(taps - 1) // 2] = 1.0
        if lo <= 0.0:
            h = lowpass(hi)
        elif hi >= 0.5 * fs:
            h = delta - lowpass(lo)                       # spectral inversion
        else:
            h = (delta - lowpass(lo)) - (delta - lowpass(hi))
        assert np.abs(filters[i] - h).max() < 1e-7, i
        assert filters[i][0] == 0.0 and filters[i][-1] == 0.0
    assert np.abs(filters.sum(0) - np.eye(1, taps, (taps - 1) // 2)[0]).max() < 1e-6   # the bank sums to a delta


def test_fir_same_slice_against_numpy():
    rng = np.random.default_rng(3)
    nt, taps = 300, 499
    x = rng.standard_normal((2, 3, nt)).astype(np.float32)
    h = (rng.standard_normal(taps) * np.hanning(taps)).astype(np.float32)
    y = ob.filter_scan(x, h)
    shift = (taps - 1) // 2
    ref = np.stack([[np.convolve(x[i, j].astype(np.float64), h.astype(np.float64))[shift:shift + nt]
                     for j in range(3)] for i in range(2)])
    assert np.abs(y - ref).max() / np.abs(ref).max() < 2e-6


@pytest.mark.parametrize("shape,pshape", [((20, 23), (7, 9)), ((18, 18), (15, 17)), ((24, 21), (17, 19)),
                                          ((30, 26), (5, 3))])
def test_richardson_lucy_against_scipy(shape, pshape):
    rng = np.random.default_rng(shape[0] * 100 + pshape[1])
    d = (0.2 + rng.random(shape)).astype(np.float32)
    yy, xx = np.mgrid[:pshape[0], :pshape[1]]
    # an asymmetric kernel: correlation and convolution differ
    psf = np.exp(-((yy - pshape[0] / 2 + 0.8) ** 2 / 6.0 + (xx - pshape[1] / 2 - 0.6) ** 2 / 9.0)).astype(np.float32)
    psf /= psf.max()
    n_iter = 4
    got = ob.richardson_lucy(d, psf, n_iter)
    small = psf.size <= 256
    conv = (lambda a, k: signal.correlate2d(a, k, mode="same")) if small else \
           (lambda a, k: signal.convolve2d(a, k, mode="same"))
    # the image is reflect-padded by half the kernel on every side, iterated on as a whole (zeros beyond the
    # padding) and cropped back (deconvolution.rs:620-712)
    py, px = pshape[0] // 2, pshape[1] // 2
    d64 = np.pad(d.astype(np.float64), ((py, py), (px, px)), mode="reflect")
    p64 = psf.astype(np.float64)

    def iterate(cv):
        u = d64.copy()
        for _ in range(n_iter):
            t = d64 / (cv(u, p64) + 1e-12)
            u = u * cv(t, p64[::-1, ::-1])
        return u[py:py + shape[0], px:px + shape[1]]

    u = iterate(conv)
    assert np.abs(got - u).max() / np.abs(u).max() < 2e-5
    # and the other convention is measurably different, i.e. the test can tell them apart
    other = (lambda a, k: signal.convolve2d(a, k, mode="same")) if small else \
            (lambda a, k: signal.correlate2d(a, k, mode="same"))
    v = iterate(other)
    assert np.abs(got - v).max() / np.abs(v).max() > 1e-3
