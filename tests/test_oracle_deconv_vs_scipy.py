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


def test_whole_deconvolution_against_a_numpy_model():
    """Deconvolution::filter end to end (deconvolution.rs:766-1041) as a float64 model: FIR bank -> per band the
    'same'-filtered cube, its energy image, Richardson-Lucy with the band's PSF and iteration count, gain
    sqrt(max(u, 0) / d) -> sum of gain-weighted filtered cubes -> intensity image.  The bank and the band PSFs are
    taken from the (separately pinned) oracle functions, everything else is numpy / scipy."""
    import os
    nx, ny, nt = 20, 18, 128
    time, cube = synth.make_cube(nx, ny, nt)
    xx, yy = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    cube = (cube * (0.4 + 0.6 * ((xx // 3) % 2 == 0))[..., None]).astype(np.float32)
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "psf_sample.npz"))
    psf = ob.psf_from_npz(z)
    n_iter, nb, f0, f1, width, dx, dy = 6, 5, 0.4, 3.0, 0.5, 0.5, 0.5
    rc, out, img, gains, niter = ob.deconvolution(cube, time, dx, dy, psf, n_iter, nb, f0, f1, width)
    assert rc == 0 and niter.max() == n_iter and niter.min() >= 1
    filters, centers = ob.filter_bank(time, nb, f0, f1, width)
    taps = filters.shape[1]
    shift = (taps - 1) // 2
    c64 = cube.astype(np.float64)
    model = np.zeros_like(c64)
    for b in range(nb):
        h = filters[b].astype(np.float64)
        y = np.apply_along_axis(lambda v: np.convolve(v, h)[shift:shift + nt], -1, c64)
        d = (y ** 2).sum(-1)
        p = ob.band_psf(psf, centers[b], dx, dy, nx, ny).astype(np.float64)
        py, px = p.shape[0] // 2, p.shape[1] // 2
        dp = np.pad(d, ((py, py), (px, px)), mode="reflect")
        cv = (lambda a, k: signal.correlate2d(a, k, mode="same")) if p.size <= 256 else \
             (lambda a, k: signal.convolve2d(a, k, mode="same"))
        u = dp.copy()
        for _ in range(int(niter[b])):
            t = dp / (cv(u, p) + 1e-12)
            u = u * cv(t, p[::-1, ::-1])
        u = u[py:py + nx, px:px + ny]
        g = np.sqrt(np.maximum(u, 0.0) / d)
        assert np.abs(gains[b] - g).max() / np.abs(g).max() < 2e-4, b
        model += g[..., None] * y
    assert np.abs(out - model).max() / np.abs(model).max() < 2e-4
    assert np.abs(img - (model ** 2).sum(-1)).max() / (model ** 2).sum(-1).max() < 4e-4
