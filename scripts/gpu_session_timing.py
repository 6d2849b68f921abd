"""Developer tool: wall time of thz_session_recompute (the whole UpdateType::Filter walk incl. host-side
multipliers, uploads and the pixel means) on a device-generated cube, the default chain."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np
import thz_image_explorer_amd as pkg
import synth
nx, ny, nt = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (1024, 1024, 4096)))
eng = pkg.Engine(0)
tm = synth.make_time(nt)
sess = pkg.Session(eng, nx, ny, tm)
raw = eng.lib.thz_session_buffer(sess.h, pkg.BUF_RAW)
d_t = eng.to_device(tm)
eng._check(eng.lib.thz_synth_cube(eng.ctx, raw, nx * ny, 0, d_t.ptr, 0x7A3D2026, 1))
eng.sync()
sess.eng._check(eng.lib.thz_session_upload(sess.h, None, 0))   # image of the resident cube, no copy
cfg = pkg.chain_cfg_default(tm)
for means in (1, 0):
    cfg.want_means = means
    sess.recompute(cfg)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); sess.recompute(cfg); ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[3]
    print(f"thz_session_recompute {nx}x{ny}x{nt} want_means={means}: {dt*1e3:.2f} ms  {nx*ny/dt/1e6:.1f} M traces/s", flush=True)
sess.close()
