"""Developer tool: wall time of thz_deconvolve with the reference's defaults (500 iterations, 25 bands,
0.1-10 THz) on a stand-in for config 4 (128 x 128 x 1001 bar-target cube, sample_data/psf.npz), and the
oracle's time for the same call with fewer iterations for scale."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import thz_image_explorer_amd as pkg
import oracle_binding as ob
from test_gpu_deconv import _bar_target_cube
nx, ny, nt = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (128, 128, 1001)))
z = np.load(os.path.join(ROOT, "tests", "golden", "psf_sample.npz"))
psf = pkg.psf_from_npz(z)
tm, cube = _bar_target_cube(nx, ny, nt)
eng = pkg.Engine(0); eng.set_time_axis(tm)
d_in = eng.to_device(cube); d_out = eng.empty((nx * ny, nt)); d_img = eng.empty((nx * ny,))
for n_iter in (500, 100):
    cfg = pkg.DeconvCfg(n_iter, 25, 0.1, 10.0, 0.5)
    eng.deconvolve(psf, cfg, nx, ny, 0.5, 0.5, d_in, d_out, d_img); eng.sync()
    t0 = time.perf_counter(); st = eng.deconvolve(psf, cfg, nx, ny, 0.5, 0.5, d_in, d_out, d_img); eng.sync()
    print(f"GPU  {nx}x{ny}x{nt} n_iterations={n_iter:4d} n_filters=25: {1e3 * (time.perf_counter() - t0):9.1f} ms (status {st})", flush=True)
if os.environ.get("THZ_DECONV_CPU"):
    opsf = ob.psf_from_npz(z)
    t0 = time.perf_counter()
    ob.deconvolution(cube, tm, 0.5, 0.5, opsf, 100, 25, 0.1, 10.0, 0.5)
    print(f"CPU oracle ({ob.max_threads()} threads) n_iterations= 100: {1e3 * (time.perf_counter() - t0):9.1f} ms", flush=True)
