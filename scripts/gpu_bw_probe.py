"""HBM streaming ceilings on the box (developer tool): d2d memcpy and the
elementwise window kernel on a 8 GiB buffer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from thz_image_explorer_amd import Engine
eng = Engine(0)
nt = 4096
eng.set_time_axis((1000 + 0.05 * np.arange(nt)).astype(np.float32))
npix = 512 * 1024
a = eng.empty((npix, nt)); b = eng.empty((npix, nt)); w = eng.to_device(np.ones(nt, np.float32))
eng.lib.thz_memset(eng.ctx, a.ptr, 0, a.nbytes); eng.sync()
for name, fn, byts in [("memcpy d2d", lambda: eng.lib.thz_memcpy_d2d(eng.ctx, b.ptr, a.ptr, a.nbytes), 2 * a.nbytes),
                       ("memset", lambda: eng.lib.thz_memset(eng.ctx, b.ptr, 0, b.nbytes), b.nbytes),
                       ("k_td_window", lambda: eng.apply_td_window(npix, a, w, b), 2 * a.nbytes)
                       ]:
    fn(); eng.sync()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); fn(); eng.sync(); ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[2]
    print(f"{name:20s} {dt*1e3:8.3f} ms  {byts/dt/1e9:8.1f} GB/s  {byts/dt/8e10:5.1f}% of 8 TB/s", flush=True)
eng.close()
