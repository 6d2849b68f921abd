"""Times thz_pipeline / thz_fft on a device-generated cube (developer tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from thz_image_explorer_amd import Engine
import synth
nx, ny, nt = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (512, 1024, 4096)))
eng = Engine(0)
eng.set_kernel_family(int(os.environ.get('THZ_FAMILY', '0')))
tm = synth.make_time(nt); eng.set_time_axis(tm); nf = eng.nf
chain = synth.default_chain(tm)
npix = nx * ny
d_t = eng.to_device(tm); d_raw = eng.empty((npix, nt)); eng.synth_cube(d_raw, npix, 0, d_t)
d_pre = eng.to_device(chain["w_pre"]); d_fd = eng.to_device(chain["fd_mask"]); d_post = eng.to_device(chain["w_post"])
d_fft = eng.empty((npix, nf, 2)); d_amp = eng.empty((npix, nf)); d_ph = eng.empty((npix, nf)); d_out = eng.empty((npix, nt)); d_img = eng.empty((npix,))
d_sums = eng.empty((2 * nf,)) if os.environ.get('THZ_SUMS') else None  # THZ_SUMS=1: the pixel sums a session recompute carries
cases = [("pipeline", (lambda: eng.pipeline_ex(npix, d_raw, d_pre, d_fd, None, d_post, d_fft, d_amp, d_ph, d_out, d_img, d_sums)) if d_sums is not None
          else (lambda: eng.pipeline(npix, d_raw, d_pre, d_fd, d_post, d_fft, d_amp, d_ph, d_out, d_img)), 16 * nt + 20),
         ("fwd M_fwd", lambda: eng.fft(npix, d_raw, d_pre, None, None, d_fft, None, None, d_fd), 8 * nt + 8),
         ("fwd all", lambda: eng.fft(npix, d_raw, d_pre, None, None, d_fft, d_amp, d_ph, d_fd), 4 * nt + 16 * nf),
         ("probe", lambda: eng.traffic_probe(npix, nt, d_raw, d_fft, d_amp, d_ph, d_out), 16 * nt + 16),
         ("inv", lambda: eng.ifft(npix, d_fft, d_post, d_out, d_img), 8 * nf + 4 * nt + 4)]
only = os.environ.get('THZ_ONLY')
for name, fn, b in cases:
    if only and name != only:
        continue
    if name == 'probe' and nt % 8:
        continue
    fn(); eng.sync()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); fn(); eng.sync(); ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[len(ts) // 2]
    print(f"{os.environ.get('THZ_F_BLOCK','512'):>4s} {eng.kernel_variant()} {nx}x{ny}x{nt} {name:10s} {dt*1e3:8.3f} ms {npix/dt/1e6:8.2f} Mtr/s {npix*b/dt/1e9:8.1f} GB/s {npix*b/dt/8e10:5.1f}%", flush=True)
eng.close()
