"""The C++ host mirror (thz_image_explorer_amd/host: ScannedImageFilterData,
math_tools::{scaling,fft,ifft}, Filter plugins + registry, the stage walk of
data_thread.rs:1090-1228) on the GPU: its self-test re-runs the reference's unit
tests, and its default-chain output is compared with the oracle walking the
same stages."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import oracle_binding as ob
import synth
from thz_image_explorer_amd import io_binding as tio

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "thz_image_explorer_amd", "host_selftest")


def test_host_mirror_selftest_and_default_chain():
    assert os.path.exists(EXE), "build it: make -C thz_image_explorer_amd/host"
    nx, ny, nt = 8, 8, 1024
    time = synth.make_time(nt)
    raw = synth.make_traces(np.arange(nx * ny), nt, subtract_bias=False).reshape(nx, ny, nt)
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "cube.bin"), "wb") as f:
            np.array([nx, ny, nt], np.int32).tofile(f)
            np.array([0.5, 0.5], np.float32).tofile(f)
            time.tofile(f)
            raw.tofile(f)
        # the PSF splines of psf.npz for the live-abort test (keys of io.rs:146-166), in the order PsfArrays holds them
        z = np.load(os.path.join(ROOT, "tests", "golden", "psf_sample.npz"))
        with open(os.path.join(d, "psf.bin"), "wb") as f:
            np.array([z["wx_base_a"], z["wx_base_b"], z["wy_base_a"], z["wy_base_b"]], np.float32).ravel().tofile(f)
            for prefix, kk, vk in (("wx_corr_", "wx_corr_knots_thz", "wx_corr_values_mm"), ("wy_corr_", "wy_corr_knots_thz", "wy_corr_values_mm"),
                                   ("x0_", "x0_knots_thz", "x0_values_mm"), ("y0_", "y0_knots_thz", "y0_values_mm")):
                for key in (kk, vk, prefix + "coeff_a", prefix + "coeff_b", prefix + "coeff_c", prefix + "coeff_d"):
                    a = np.asarray(z[key], np.float64).astype(np.float32).ravel()
                    np.array([a.size], np.int32).tofile(f)
                    a.tofile(f)
        if tio.available():
            tio.save_scan(os.path.join(d, "scan.thzimg"), time, raw,
                          {"width": nx, "height": ny, "dx [mm]": "0.5", "dy [mm]": "0.25", "user": "test"})
        r = subprocess.run([EXE, d], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        print(r.stdout)
        assert r.returncode == 0, r.stdout
        assert "SELFTEST OK" in r.stdout and "FAIL" not in r.stdout
        out = np.fromfile(os.path.join(d, "out.bin"), np.float32)
        out2 = np.fromfile(os.path.join(d, "out2.bin"), np.float32)
        out3 = np.fromfile(os.path.join(d, "out3.bin"), np.float32) if tio.available() else None
    nf = nt // 2 + 1
    npix = nx * ny
    sizes = [npix * nf * 2, npix * nf, npix * nf, npix * nt, npix, nf, nf, nt]
    assert out.size == sum(sizes)
    parts = np.split(out, np.cumsum(sizes)[:-1])
    fft, amp, ph, data, img, avg_sig, avg_ph, roi = parts
    # oracle: the same stage walk (bias -> tilt taper -> Time Band Pass -> fft ->
    # Frequency Band Pass -> ifft -> Time Band Pass -> image)
    cube = ob.subtract_bias(raw)
    chain = synth.default_chain(time)
    ref = ob.run_pipeline(cube, time, chain)
    scale = np.abs(ref["fft"]).max()
    assert np.abs(fft.reshape(ref["fft"].shape) - ref["fft"]).max() / scale < 1e-5
    assert np.abs(amp.reshape(ref["amplitudes"].shape) - ref["amplitudes"]).max() / scale < 1e-5
    assert np.abs(data.reshape(ref["data"].shape) - ref["data"]).max() / np.abs(ref["data"]).max() < 1e-5
    assert np.abs(img.reshape(ref["img"].shape) - ref["img"]).max() / ref["img"].max() < 1e-5
    # K8 means (taken by `ifft` from its input = the band-passed spectra)
    assert np.abs(avg_sig - ob.pixel_mean(ref["amplitudes"], 1)).max() / scale < 1e-5
    d = avg_ph - ob.pixel_mean(ref["phases"], 1)
    assert np.abs(d).max() < 0.2  # a 2*pi flip on one noise bin of one pixel moves the mean by 2*pi/64
    # K9 ROI mean of the ifft stage's *input* data (math_tools.rs:477: &input.data)
    pre = cube * chain["w_tilt"]
    pre, _, _ = ob.td_bandpass(pre, time, float(time[0]), float(time[-1]), 2.0)
    st = ob.fft_stage(pre, time, 0, 1.0, 7.0)
    poly = np.array([[1, 1], [5, 1], [6, 4], [3, 6], [1, 4]], np.uint64)
    assert np.array_equal(roi, ob.average_polygon_roi(st["data"], poly, 1))

    # ---- second run of the self-test: scale_factor = 2, avg_in_fourier_space = true
    sx, sy = nx // 2, ny // 2
    sizes2 = [sx * sy * nt, nx * ny, nt, nf, nt]
    assert out2.size == sum(sizes2)
    data2, img2, avg_data2, avg_sig2, roi2 = np.split(out2, np.cumsum(sizes2)[:-1])
    small = ob.scale3d(cube, 2)  # math_tools::scaling (block mean, :273-301)
    ref2 = ob.run_pipeline(small, time, chain)
    assert np.abs(data2.reshape(ref2["data"].shape) - ref2["data"]).max() / np.abs(ref2["data"]).max() < 1e-5
    # data_thread.rs:1243-1285: image of the scaled data, replicated s x s
    big = np.repeat(np.repeat(ref2["img"], 2, axis=0), 2, axis=1)
    assert np.abs(img2.reshape(nx, ny) - big).max() / big.max() < 1e-5
    # math_tools.rs:442-470: avg_data = C2R(from_polar(mean |X|, mean phase)) / nt
    m_amp = ob.pixel_mean(ref2["amplitudes"], 1)
    assert np.abs(avg_sig2 - m_amp).max() / np.abs(ref2["fft"]).max() < 1e-5
    assert np.isfinite(avg_data2).all() and np.abs(avg_data2).max() > 0
    assert np.isfinite(roi2).all() and roi2.size == nt

    # ---- third run: the same cube opened from a .thzimg file, streamed in 3-row slabs
    if out3 is not None:
        raw3, img3, data3 = np.split(out3, [npix * nt, npix * nt + npix])
        assert np.array_equal(raw3.reshape(cube.shape), cube)                  # bias subtraction is exact
        assert np.abs(img3.reshape(nx, ny) - ob.intensity(cube)).max() / ob.intensity(cube).max() < 1e-5
        assert np.abs(data3.reshape(ref["data"].shape) - ref["data"]).max() / np.abs(ref["data"]).max() < 1e-5
