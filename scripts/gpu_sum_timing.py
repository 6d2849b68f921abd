"""Times thz_pixel_sum on spectrum-sized arrays (developer tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from thz_image_explorer_amd import Engine
eng = Engine(0)
nt = 4096; eng.set_time_axis((1000 + 0.05 * np.arange(nt)).astype(np.float32)); nf = eng.nf
npix = 512 * 1024
a = eng.empty((npix, nf, 2)); eng.lib.thz_memset(eng.ctx, a.ptr, 0, a.nbytes)
o = eng.empty((4 * nf,))
for name, ncomp, byts in (("fft(c32)", 2, npix * nf * 8), ("amp(f32)", 1, npix * nf * 4)):
    eng.pixel_sum(npix, nf, ncomp, a, o); eng.sync()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); eng.pixel_sum(npix, nf, ncomp, a, o); eng.sync(); ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[2]
    print(f"pixel_sum {name}: {dt*1e3:.3f} ms  {byts/dt/1e9:.0f} GB/s", flush=True)
eng.close()
