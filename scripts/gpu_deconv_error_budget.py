"""Developer tool (DESIGN.md §4.3 error budget): distance of the device's Deconvolution from the oracle as a function
of the iteration count, and how much of it the fp32 FIR explains.

The reference convolves every trace with each band's 499-tap FIR through a Complex<f64> FFT
(deconvolution.rs:266-317); the device does one fp32 transform per trace.  Richardson-Lucy then iterates on the
band energy images (up to 500 times on the widest band) and can amplify whatever the energies differ by.
Columns: relative max-norm error of the output cube / the band gains / the image
  device   vs oracle (f64 FIR, the reference's arithmetic)
  f32-FIR  oracle with its FIR switched to an f32 FFT (diagnostic knob) vs the same oracle with f64
Usage: scripts/gpu_deconv_error_budget.py [n=64]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_binding as ob
import thz_image_explorer_amd as pkg
from test_gpu_configs import resolution_target_stand_in

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
tm, cube = resolution_target_stand_in(n, n)
nt = tm.size
z = np.load(os.path.join(ROOT, "tests", "golden", "psf_sample.npz"))
psf, opsf = pkg.psf_from_npz(z), ob.psf_from_npz(z)
eng = pkg.Engine(0)
eng.set_time_axis(tm)
d_in = eng.to_device(cube); d_out = eng.empty((n * n, nt)); d_img = eng.empty((n * n,)); d_g = eng.empty((25, n * n))
rel = lambda a, b: float(np.abs(a.astype(np.float64) - b).max() / np.abs(b).max())
print(f"{n}x{n}x{nt} bar-target stand-in, psf.npz, 25 bands 0.1-10 THz, dx = dy = 0.5 mm")
print(f"{'iterations':>10s} | {'device: cube':>12s} {'gains':>9s} {'image':>9s} | {'f32-FIR oracle: cube':>20s} {'gains':>9s} {'image':>9s} | widest band iterations")
for it in ((1, 5, 20, 100, 500) if os.environ.get('THZ_BUDGET_TABLE', '1') == '1' else ()):
    cfg = pkg.DeconvCfg(it, 25, 0.1, 10.0, 0.5)
    assert eng.deconvolve(psf, cfg, n, n, 0.5, 0.5, d_in, d_out, d_img, d_g) == 0
    out = d_out.download((n, n, nt), np.float32); img = d_img.download((n, n), np.float32); g = d_g.download((25, n, n), np.float32)
    ob.lib().thz_oracle_set_fir_f32(0)
    rc, oref, oimg, og, onit = ob.deconvolution(cube, tm, 0.5, 0.5, opsf, it, 25, 0.1, 10.0, 0.5)
    ob.lib().thz_oracle_set_fir_f32(1)
    rc2, fref, fimg, fg, _ = ob.deconvolution(cube, tm, 0.5, 0.5, opsf, it, 25, 0.1, 10.0, 0.5)
    ob.lib().thz_oracle_set_fir_f32(0)
    print(f"{it:10d} | {rel(out, oref):12.2e} {rel(g, og):9.2e} {rel(img, oimg):9.2e} | {rel(fref, oref):20.2e} {rel(fg, og):9.2e} {rel(fimg, oimg):9.2e} | {int(onit.max())}", flush=True)
eng.close()


# ---- the cases of tests/test_gpu_deconv.py::test_deconvolution_vs_oracle and config 4, for the bars asserted there
import synth
from test_gpu_deconv import _bar_target_cube
eng = pkg.Engine(0)
print("\ncases of tests/test_gpu_deconv.py (nx, ny, nt, dx, iterations, bands):  cube  gains  image")
for case in (dict(nx=32, ny=32, nt=256, d=0.5, n_iter=20, nf=6, f0=0.4, f1=3.0),
             dict(nx=48, ny=40, nt=128, d=1.0, n_iter=12, nf=4, f0=0.25, f1=2.0),
             dict(nx=20, ny=18, nt=1001, d=0.5, n_iter=6, nf=5, f0=0.4, f1=3.0),
             dict(nx=18, ny=20, nt=2000, d=0.5, n_iter=6, nf=4, f0=0.4, f1=3.0),
             dict(nx=16, ny=17, nt=4000, d=0.5, n_iter=4, nf=3, f0=0.4, f1=3.0),
             dict(nx=48, ny=40, nt=128, d=1.0, n_iter=200, nf=4, f0=0.25, f1=2.0)):
    tm2, cube2 = _bar_target_cube(case["nx"], case["ny"], case["nt"])
    eng.set_time_axis(tm2)
    n2 = case["nx"] * case["ny"]
    di = eng.to_device(cube2); do = eng.empty((n2, case["nt"])); dim = eng.empty((n2,)); dg = eng.empty((case["nf"], n2))
    cfg = pkg.DeconvCfg(case["n_iter"], case["nf"], case["f0"], case["f1"], 0.5)
    assert eng.deconvolve(psf, cfg, case["nx"], case["ny"], case["d"], case["d"], di, do, dim, dg) == 0
    rc, oref, oimg, og, onit = ob.deconvolution(cube2, tm2, case["d"], case["d"], opsf, case["n_iter"], case["nf"], case["f0"], case["f1"], 0.5)
    print(f"  {case['nx']:3d} {case['ny']:3d} {case['nt']:5d} {case['d']:.1f} {case['n_iter']:4d} {case['nf']:2d}:  "
          f"{rel(do.download((case['nx'], case['ny'], case['nt']), np.float32), oref):.2e}  "
          f"{rel(dg.download((case['nf'], case['nx'], case['ny']), np.float32), og):.2e}  "
          f"{rel(dim.download((case['nx'], case['ny']), np.float32), oimg):.2e}", flush=True)
    for b in (di, do, dim, dg):
        b.free()
eng.close()
