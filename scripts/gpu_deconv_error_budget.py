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
for it in (1, 5, 20, 100, 500):
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
