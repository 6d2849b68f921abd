"""Regions of interest and avg_in_fourier_space on the session / group fast path (thz_session_set_rois,
thz_session_roi, thz_group_session_*): what the reference's ifft stage computes per region with every recompute
(math_tools.rs:473-543, average_polygon_roi :594-661) and what the plot copy-out reads (data_thread.rs:1442-1482).

Bars: masks and counts exact (integer rule); with want_means == 2 every mean is the reference-order mean of the
RESIDENT array bit for bit (the oracle's average_polygon_roi applied to the downloaded array), and roi_data — the mean
of the fft stage's windowed traces, which the device reproduces bit for bit — equals the oracle's outright; against
the oracle's own arrays 1e-5; the parallel sums of want_means 0 / 1 and the slabs of a group 2e-6."""
import os

import numpy as np
import pytest

import oracle_binding as ob
import synth
import thz_image_explorer_amd as pkg
from test_gpu_parity import TOL, rel
from test_gpu_session import oracle_chain

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def polygons():
    g = np.load(os.path.join(GOLD, "roi_masks.npz"))
    return {k[: -len("_poly")]: g[k].astype(np.uint64) for k in g.files if k.endswith("_poly")}


def windowed_input(cube, time, cfg, dx=1.0, dy=1.0):
    """the fft stage's `data` output (= the ifft stage's input.data): oracle, stage by stage"""
    d, t = cube, time
    if cfg.tilt_active:
        _, t, d = ob.tilt(d, t, cfg.tilt_x_deg, cfg.tilt_y_deg, dx, dy)
    if cfg.td_before_active:
        d, _, _ = ob.td_bandpass(d, t, cfg.td_before_low, cfg.td_before_high, cfg.td_before_width)
    return ob.fft_stage(d, t, cfg.fft_window.type, cfg.fft_window.lower, cfg.fft_window.upper)["data"]


def near(a, b, tol):
    return np.abs(a.astype(np.float64) - b).max() <= tol * max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("shape,scale", [((32, 32, 256), 1), ((24, 40, 1001), 1), ((32, 32, 1024), 2)])
def test_session_roi_means_reference_order(engine, shape, scale):
    nx, ny, nt = shape
    time, cube = synth.make_cube(nx, ny, nt)
    polys = polygons()
    names = sorted(polys)
    sess = pkg.Session(engine, nx, ny, time)
    try:
        sess.upload(cube, subtract_bias=False)
        sess.set_rois([polys[n] for n in names])
        cfg = pkg.chain_cfg_default(time)
        cfg.want_means = 2
        cfg.scale_factor = scale
        with pytest.raises(pkg.ThzError):
            sess.roi(0)                       # nothing computed since the regions were set
        sess.recompute(cfg)
        gx, gy = sess.grid()[:2]
        nf = nt // 2 + 1
        amp = sess.download(pkg.BUF_AMPLITUDES).reshape(gx, gy, nf)
        ph = sess.download(pkg.BUF_PHASES).reshape(gx, gy, nf)
        data = sess.download(pkg.BUF_DATA).reshape(gx, gy, nt)
        src = ob.scale3d(cube, scale) if scale > 1 else cube
        ref = oracle_chain(src, time, cfg)
        win = windowed_input(src, time, cfg)
        for i, name in enumerate(names):
            poly = polys[name]
            r = sess.roi(i)
            mask, _ = ob.roi_mask(poly, scale, gx, gy)
            assert r["count"] == int(mask.sum()), name
            # reference-order means of the resident arrays, bit for bit
            assert np.array_equal(r["signal_fft"], ob.average_polygon_roi(amp, poly, scale)), name
            assert np.array_equal(r["phase_fft"], ob.average_polygon_roi(ph, poly, scale)), name
            assert np.array_equal(r["signal"], ob.average_polygon_roi(data, poly, scale)), name
            # the ifft stage's roi_data: mean of the windowed input traces — the oracle's own number
            assert np.array_equal(r["roi_data"], ob.average_polygon_roi(win, poly, scale)), name
            # ... and against the oracle's arrays
            if mask.any():
                assert near(r["signal_fft"], ob.average_polygon_roi(ref["amp"], poly, scale), TOL), name
                assert near(r["signal"], ob.average_polygon_roi(ref["data"], poly, scale), TOL), name
            else:
                assert not r["signal_fft"].any() and not r["signal"].any() and not r["roi_data"].any()
        # UpdateType::Filter(7): only the final traces change; the regions' amplitude / phase means stay
        before = sess.roi(0)
        cfg.td_after_high = float(time[-1]) - 3.0
        sess.recompute(cfg, start_stage=7)
        data2 = sess.download(pkg.BUF_DATA).reshape(gx, gy, nt)
        after = sess.roi(0)
        assert np.array_equal(after["signal"], ob.average_polygon_roi(data2, polys[names[0]], scale))
        assert np.array_equal(after["signal_fft"], before["signal_fft"]) and np.array_equal(after["roi_data"], before["roi_data"])
        assert not np.array_equal(after["signal"], before["signal"])
        # new regions: every sum afresh, whatever the start position of the next recompute
        sess.set_rois([polys["triangle"]])
        sess.recompute(cfg, start_stage=7)
        assert np.array_equal(sess.roi(0)["signal_fft"], ob.average_polygon_roi(amp, polys["triangle"], scale))
        sess.set_rois([])
        sess.recompute(cfg)
        with pytest.raises(pkg.ThzError):
            sess.roi(0)
    finally:
        sess.close()


@pytest.mark.parametrize("shape", [(32, 32, 1024), (129, 257, 256), (24, 40, 1001)])
def test_session_roi_means_parallel_sums(engine, shape):
    """want_means 0 / 1: the regions' rows are added in parallel (like the pixel sums)"""
    nx, ny, nt = shape
    time, cube = synth.make_cube(nx, ny, nt)
    polys = polygons()
    names = sorted(polys)
    sess = pkg.Session(engine, nx, ny, time)
    try:
        sess.upload(cube, subtract_bias=False)
        sess.set_rois([polys[n] for n in names])
        cfg = pkg.chain_cfg_default(time)
        sess.recompute(cfg)
        nf = nt // 2 + 1
        amp = sess.download(pkg.BUF_AMPLITUDES).reshape(nx, ny, nf)
        ph = sess.download(pkg.BUF_PHASES).reshape(nx, ny, nf)
        data = sess.download(pkg.BUF_DATA).reshape(nx, ny, nt)
        win = windowed_input(cube, time, cfg)
        for i, name in enumerate(names):
            r = sess.roi(i)
            mask, _ = ob.roi_mask(polys[name], 1, nx, ny)
            assert r["count"] == int(mask.sum())
            sel = np.flipud(mask).astype(bool)           # mask position (x, y) samples pixel [shape0 - y - 1, x]
            for key, arr in (("signal_fft", amp), ("phase_fft", ph), ("signal", data), ("roi_data", win)):
                want = arr[sel].astype(np.float64).mean(0) if sel.any() else np.zeros(arr.shape[2])
                assert near(r[key], want, 2e-6), (name, key)
    finally:
        sess.close()


@pytest.mark.parametrize("nt,bandpass", [(1024, True), (1001, True), (1024, False), (1001, False)])
def test_avg_in_fourier_space(engine, nt, bandpass):
    """ConfigContainer.avg_in_fourier_space: avg_data and the regions' roi_data from the polar form
    (math_tools.rs:442-470, 496-529); without the band pass an even length leaves an imaginary part in the last bin,
    realfft refuses it and the reference falls back to the traces (:530-538)"""
    nx, ny = 16, 20
    time, cube = synth.make_cube(nx, ny, nt)
    poly = np.array([[2, 3], [15, 2], [17, 12], [8, 14], [1, 9]], np.uint64)
    sess = pkg.Session(engine, nx, ny, time)
    try:
        sess.upload(cube, subtract_bias=False)
        sess.set_rois([poly])
        cfg = pkg.chain_cfg_default(time)
        cfg.want_means = 2
        cfg.avg_in_fourier_space = 1
        cfg.fd_active = 1 if bandpass else 0
        sess.recompute(cfg)
        r = sess.roi(0)
        amp_m, ph_m = r["signal_fft"], r["phase_fft"]
        refused = nt % 2 == 0 and np.float32(amp_m[-1] * np.sin(ph_m[-1], dtype=np.float32)) != 0
        assert refused == (nt % 2 == 0 and not bandpass)
        if refused:
            win = windowed_input(cube, time, cfg)
            assert np.array_equal(r["roi_data"], ob.average_polygon_roi(win, poly, 1))
        else:
            want = ob.polar_irfft(amp_m, ph_m, nt, zero_dc_imag=True)
            assert near(r["roi_data"], want, TOL)
        assert np.array_equal(r["signal"], r["roi_data"])             # data_thread.rs:1476-1482
        avg_a = sess.download(pkg.BUF_AVG_AMPLITUDES); avg_p = sess.download(pkg.BUF_AVG_PHASES)
        got = sess.plot(0, 0, want=["avg_signal"])["avg_signal"]
        assert near(got, ob.polar_irfft(avg_a, avg_p, nt, zero_dc_imag=False), TOL)
        # back to averaging in time: the final cube's pixel mean again
        cfg.avg_in_fourier_space = 0
        sess.recompute(cfg)
        data = sess.download(pkg.BUF_DATA).reshape(nx, ny, nt)
        assert np.array_equal(sess.plot(0, 0, want=["avg_signal"])["avg_signal"], ob.pixel_mean(data))
        assert np.array_equal(sess.roi(0)["signal"], ob.average_polygon_roi(data, poly, 1))
    finally:
        sess.close()


@pytest.mark.parametrize("shape,members", [((32, 32, 1024), 2), ((33, 32, 1001), 3), ((129, 40, 256), 4)])
def test_group_roi_equals_single_session(engine, shape, members):
    """every slab sums its rows of the whole grid's mask (the flipped row index runs along the sharded axis), C2
    all-reduces the sums: == one session over the whole cube"""
    nx, ny, nt = shape
    time, cube = synth.make_cube(nx, ny, nt)
    polys = polygons()
    names = [n for n in sorted(polys)]
    cfg = pkg.chain_cfg_default(time)
    single = pkg.Session(engine, nx, ny, time)
    try:
        single.upload(cube, subtract_bias=False)
        single.set_rois([polys[n] for n in names])
        single.recompute(cfg)
        want = [single.roi(i) for i in range(len(names))]
        cfg7 = pkg.chain_cfg_default(time)
        cfg7.td_after_high = float(time[-1]) - 3.0
        single.recompute(cfg7, start_stage=7)
        want7 = [single.roi(i) for i in range(len(names))]
    finally:
        single.close()
    with pkg.Group(devices=[0] * members) as g:
        gs = pkg.GroupSession(g, nx, ny, time)
        try:
            gs.upload(cube, subtract_bias=False)
            gs.set_rois([polys[n] for n in names])
            gs.recompute(cfg, 1, pkg.GATHER_SMALL)
            for i, name in enumerate(names):
                r = gs.roi(i)
                assert r["count"] == want[i]["count"], name
                for key in ("signal_fft", "phase_fft", "signal", "roi_data"):
                    assert near(r[key], want[i][key].astype(np.float64), 2e-6), (name, key)
            gs.recompute(cfg7, 7, pkg.GATHER_SMALL)      # the chain's tail: only the final traces' block is renewed
            for i, name in enumerate(names):
                r = gs.roi(i)
                for key in ("signal_fft", "phase_fft", "signal", "roi_data"):
                    assert near(r[key], want7[i][key].astype(np.float64), 2e-6), (name, key)
        finally:
            gs.close()
