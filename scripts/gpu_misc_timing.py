"""Developer tool: achieved HBM rates of the elementwise / reduction stage kernels on a large cube."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import thz_image_explorer_amd as pkg
import synth
nx, ny, nt = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (512, 512, 4096)))
eng = pkg.Engine(0)
tm = synth.make_time(nt); eng.set_time_axis(tm); nf = eng.nf
npix = nx * ny
d_t = eng.to_device(tm); d_a = eng.empty((npix, nt)); eng.synth_cube(d_a, npix, 0, d_t)
d_b = eng.empty((npix, nt)); d_img = eng.empty((npix,))
d_w = eng.to_device(np.linspace(0.5, 1.5, nt).astype(np.float32))
d_fft = eng.empty((npix, nf, 2)); d_amp = eng.empty((npix, nf)); d_m = eng.to_device(np.linspace(0, 1, nf).astype(np.float32))
d_small = eng.empty((npix // 4, nt))
steps, new_time, ins = pkg.host_tilt_plan(tm, nx, ny, 1.0, 0.5, 0.5, 0.5)
nt2 = nt + 2 * steps
d_ins = eng.to_device(ins); d_tilt = eng.empty((npix, nt2))
mask = np.zeros((nx, ny), np.uint8)
poly = np.array([[nx // 8, ny // 8], [7 * nx // 8, ny // 6], [6 * nx // 8, 5 * ny // 6], [nx // 3, 7 * ny // 8]], np.uint64)
d_mask = eng.alloc(nx * ny); d_roi = eng.empty((nt,)); d_cnt = eng.alloc(4)
eng.roi_mask(poly, 1, nx, ny, d_mask)
import time
def t(name, fn, nbytes):
    fn(); eng.sync()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); fn(); eng.sync(); ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[2]
    print(f"{name:34s} {dt*1e3:8.3f} ms {nbytes/dt/1e9:8.1f} GB/s {nbytes/dt/8e10:5.1f} %", flush=True)
N = npix * nt * 4
t("td_window (read+write)", lambda: eng.apply_td_window(npix, d_a, d_w, d_b), 2 * N)
t("intensity (read)", lambda: eng.intensity(npix, d_a, d_img), N)
t("subtract_bias (read+write)", lambda: eng.subtract_bias(npix, d_b, d_img), 2 * N)
t("fd_mask (fft+amp r+w)", lambda: eng.apply_fd_mask(npix, d_fft, d_amp, d_m), 2 * npix * nf * 12)
t("scale3d s=2 (read + 1/4 write)", lambda: eng.scale3d(d_a, nx, ny, nt, 1, 2, d_small), N + N // 4)
t(f"tilt_apply nt {nt}->{nt2}", lambda: eng.tilt_apply(npix, d_a, nt, d_w, d_ins, nt2, d_tilt), N + npix * nt2 * 4)
t("pixel_mean (read)", lambda: eng.pixel_mean(nx, ny, nt, 1, d_a, d_roi), N)
cnt = int(np.unpackbits(d_mask.download((nx * ny,), np.uint8) > 0).sum()) if False else int((d_mask.download((nx * ny,), np.uint8) > 0).sum())
t(f"roi_mean ({cnt} px inside)", lambda: eng.roi_mean(d_a, nx, ny, nt, d_mask, d_roi, d_cnt), cnt * nt * 4)
eng.enable_timing(1); eng.roi_mean(d_a, nx, ny, nt, d_mask, d_roi, d_cnt)
ns = eng.stage_time_ns(pkg.binding.STAGE_ROI)
print(f"  roi_mean kernel alone: {ns/1e6:.3f} ms  {cnt*nt*4/ns:.1f} GB/s")
eng.close()
