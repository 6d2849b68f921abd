"""The record-and-flush engine behind the reference-side binding (thz_image_explorer_amd/host/thz_engine.*: the C++
twin of rust/engine.rs, rust/math_tools_gpu.rs, rust/filters/*.rs and rust/data_thread.patch) driven through the
patched stage walk the way the reference's data thread would drive it — filters switched off and on, one slider per
chain position, the Frequency plugins, regions of interest, scaling, the Deconvolution stage on two slabs — and every
dump compared with the oracle walking the same chain."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import oracle_binding as ob
import synth
import thz_image_explorer_amd as pkg
from test_gpu_parity import TOL, rel
from test_gpu_session import oracle_chain

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "thz_image_explorer_amd", "engine_selftest")


def read_dump(path, nt_in):
    raw = np.fromfile(path, np.uint8)
    nto, gx, gy, nroi = np.frombuffer(raw[:16].tobytes(), np.int32)
    f = np.frombuffer(raw[16:].tobytes(), np.float32)
    nf = nto // 2 + 1
    o = 0
    cube = f[o:o + gx * gy * nto].reshape(gx, gy, nto); o += gx * gy * nto
    n_img = f.size - o - nf - nroi * (2 * nf + nto)
    img = f[o:o + n_img]; o += n_img
    avg_amp = f[o:o + nf]; o += nf
    rois = []
    for _ in range(nroi):
        rois.append(dict(amp=f[o:o + nf], ph=f[o + nf:o + 2 * nf], data=f[o + 2 * nf:o + 2 * nf + nto]))
        o += 2 * nf + nto
    assert o == f.size
    return dict(nt_out=int(nto), cube=cube, img=img, avg_amp=avg_amp, rois=rois)


def test_engine_walks_like_the_data_thread():
    assert os.path.exists(EXE), "build it: make -C thz_image_explorer_amd/host"
    nx, ny, nt = 36, 32, 256
    time = synth.make_time(nt)
    raw = synth.make_traces(np.arange(nx * ny), nt, subtract_bias=False).reshape(nx, ny, nt)
    lines = np.loadtxt(os.path.join(ROOT, "tests", "golden", "water_lines.csv"), dtype=np.float32).ravel()
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "cube.bin"), "wb") as f:
            np.array([nx, ny, nt], np.int32).tofile(f)
            np.array([0.5, 0.5], np.float32).tofile(f)
            time.tofile(f)
            raw.tofile(f)
        with open(os.path.join(d, "water_lines.bin"), "wb") as f:
            np.array([lines.size], np.int32).tofile(f)
            lines.tofile(f)
        z = np.load(os.path.join(ROOT, "tests", "golden", "psf_sample.npz"))
        with open(os.path.join(d, "psf.bin"), "wb") as f:
            np.array([z["wx_base_a"], z["wx_base_b"], z["wy_base_a"], z["wy_base_b"]], np.float32).ravel().tofile(f)
            for prefix, kk, vk in (("wx_corr_", "wx_corr_knots_thz", "wx_corr_values_mm"), ("wy_corr_", "wy_corr_knots_thz", "wy_corr_values_mm"),
                                   ("x0_", "x0_knots_thz", "x0_values_mm"), ("y0_", "y0_knots_thz", "y0_values_mm")):
                for key in (kk, vk, prefix + "coeff_a", prefix + "coeff_b", prefix + "coeff_c", prefix + "coeff_d"):
                    a = np.asarray(z[key], np.float64).astype(np.float32).ravel()
                    np.array([a.size], np.int32).tofile(f)
                    a.tofile(f)
        r = subprocess.run([EXE, d], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        print(r.stdout)
        assert r.returncode == 0, r.stdout
        assert "ENGINE SELFTEST OK" in r.stdout and "FAIL" not in r.stdout
        dumps = {n[:-4]: read_dump(os.path.join(d, n), nt) for n in os.listdir(d)
                 if n.endswith(".bin") and n not in ("cube.bin", "psf.bin", "water_lines.bin")}

    cube = ob.subtract_bias(raw)

    def check(name, cfg, src=None, scale=1):
        got = dumps[name]
        ref = oracle_chain(cube if src is None else src, time, cfg, 0.5 * scale, 0.5 * scale)
        assert got["nt_out"] == ref["time"].size, name
        assert rel(got["cube"], ref["data"]) < TOL, name
        img = ref["img"] if scale == 1 else np.repeat(np.repeat(ref["img"], scale, axis=0), scale, axis=1)
        assert rel(got["img"].reshape(img.shape), img) < TOL, name
        assert rel(got["avg_amp"], ref["avg"]["amp"]) < TOL, name
        return ref

    cfg = pkg.chain_cfg_default(time)
    check("default", cfg)
    # each plugin switched off: the oracle's walk clones its input there (data_thread.rs:1185-1188)
    c = pkg.chain_cfg_default(time); c.fd_active = 0
    check("fd_off", c)
    c = pkg.chain_cfg_default(time); c.td_before_active = 0
    check("tdb_off", c)
    c = pkg.chain_cfg_default(time); c.td_after_active = 0
    check("tda_off", c)
    c = pkg.chain_cfg_default(time); c.tilt_active = 0
    check("tilt_off", c)
    # one slider per position
    s = pkg.chain_cfg_default(time)
    s.fd_low, s.fd_high = 0.4, 2.5
    s.td_after_high = float(time[-1]) - 4.0
    s.fft_window.lower, s.fft_window.upper = 0.5, 3.0
    ref_s = check("sliders", s)
    # K14 switched on: band-pass mask times the notch mask, numpy fp64 on the oracle's vectors
    freq = ob.frequency_axis(time).astype(np.float64)
    notch = np.prod(1.0 - np.exp(-((freq[:, None] - lines[None, :].astype(np.float64)) / 0.02) ** 2), axis=1)
    ch = synth.oracle_chain(time)
    pre = (ch["w_tilt"].astype(np.float64) * ch["w_td_before"] * ob.apply_window(0, np.ones(nt, np.float32), time, 0.5, 3.0))
    X = np.fft.rfft(cube.reshape(-1, nt).astype(np.float64) * pre, axis=1)
    m = ob.fd_bandpass_window(ob.frequency_axis(time), 0.4, 2.5, 0.1)[0].astype(np.float64) * notch
    post = ob.td_bandpass_window(time, float(time[0]), float(time[-1]) - 4.0, 0.1)[0]
    want = np.fft.irfft(X * m, n=nt, axis=1) * post
    assert rel(dumps["water_on"]["cube"].reshape(-1, nt), want) < TOL
    assert rel(dumps["water_on"]["cube"], ref_s["data"]) > 1e-3            # the notch did something
    assert np.array_equal(dumps["water_off"]["cube"], dumps["sliders"]["cube"])   # and is gone again
    # regions of interest through the ifft stage (map order: roi-a, roi-b; the region without a polygon has no entry)
    polys = [np.array([[1, 1], [5, 1], [6, 4], [3, 6], [1, 4]], np.uint64), np.array([[8, 2], [14, 3], [12, 12]], np.uint64)]
    got = dumps["rois"]
    assert len(got["rois"]) == 2
    ref_r = oracle_chain(cube, time, s, 0.5, 0.5)
    for g, poly in zip(got["rois"], polys):
        assert rel(g["amp"], ob.average_polygon_roi(ref_r["amp"], poly, 1)) < TOL
        d = g["ph"] - ob.average_polygon_roi(ref_r["ph"], poly, 1)
        assert np.abs(d).max() < 0.5      # a 2 pi flip on one noise bin of one pixel moves a region's mean by 2 pi / count
    gf = dumps["rois_fourier"]
    for g in gf["rois"]:
        assert rel(g["data"], ob.polar_irfft(g["amp"], g["ph"], nt, zero_dc_imag=True)) < TOL   # math_tools.rs:496-529
    # scaling: SetDownScaling -> Filter(1)
    check("scaled", s, src=ob.scale3d(cube, 2), scale=2)   # (reset() is not called again: every slider stays where it was)
    # Deconvolution: group == single is asserted by the self-test; against the oracle here
    psf = ob.psf_from_npz(z)
    before = ref_r["data"]
    rc, dec, *_ = ob.deconvolution(before, time, 0.5, 0.5, psf, 20, 5, 0.4, 3.0, 0.5)
    assert rc == 0
    assert rel(dumps["deconv_single"]["cube"], dec) < 3e-5
    assert rel(dumps["deconv_group"]["cube"], dec) < 3e-5
    assert rel(dumps["deconv_off"]["cube"], before) < TOL
