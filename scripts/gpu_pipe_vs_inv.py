"""Developer check: the fused chain's time-domain output against the stand-alone inverse of the spectrum it stored,
and both against numpy fp64 (same spectrum, same window)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from thz_image_explorer_amd import Engine
import synth
for nt in (1001, 1000, 1024, 1500):
    nx, ny = 8, 8
    time, cube = synth.make_cube(nx, ny, nt)
    eng = Engine(0); eng.set_time_axis(time); nf = eng.nf; npix = nx * ny
    chain = synth.default_chain(time)
    d_raw = eng.to_device(cube); d_pre = eng.to_device(chain["w_pre"]); d_fd = eng.to_device(chain["fd_mask"]); d_post = eng.to_device(chain["w_post"])
    d_fft = eng.empty((npix, nf, 2)); d_amp = eng.empty((npix, nf)); d_ph = eng.empty((npix, nf)); d_out = eng.empty((npix, nt)); d_out2 = eng.empty((npix, nt)); d_img = eng.empty((npix,))
    eng.pipeline(npix, d_raw, d_pre, d_fd, d_post, d_fft, d_amp, d_ph, d_out, d_img); eng.sync()
    eng.ifft(npix, d_fft, d_post, d_out2, d_img); eng.sync()
    a = d_out.download((npix, nt), np.float32); b = d_out2.download((npix, nt), np.float32)
    f = d_fft.download((npix, nf, 2), np.float32).astype(np.float64)
    Y = f[..., 0] + 1j * f[..., 1]
    ref = np.fft.irfft(Y, n=nt, axis=1) * chain["w_post"].astype(np.float64)
    s = np.abs(ref).max()
    print(f"nt {nt} {eng.kernel_variant()}: fused vs inverse max |diff| / max {np.abs(a - b).max() / s:.2e} (equal: {np.array_equal(a, b)}); fused vs fp64 {np.abs(a - ref).max() / s:.2e}; inverse vs fp64 {np.abs(b - ref).max() / s:.2e}", flush=True)
    eng.close()
