"""Developer tool: wall time of thz_session_recompute_from per chain position (UpdateType::Filter(start_idx)) on the
BASELINE cube, beside the pieces it is made of.  Usage: scripts/gpu_session_timing.py [nx ny nt]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import thz_image_explorer_amd as pkg
import synth

nx, ny, nt = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (1024, 1024, 4096)))
eng = pkg.Engine(0)
tm = synth.make_time(nt)
sess = pkg.Session(eng, nx, ny, tm)
d_t = eng.to_device(tm)
eng.synth_cube(eng.lib.thz_session_buffer(sess.h, pkg.BUF_RAW), nx * ny, 0, d_t)
t0 = time.perf_counter(); sess.upload(None, subtract_bias=False); t_up = time.perf_counter() - t0
cfg = pkg.chain_cfg_default(tm)
names = {1: "scaling (everything)", 3: "Time Band Pass / fft window", 5: "Frequency Band Pass", 6: "ifft",
         7: "Time Band Pass (after)"}
print(f"{nx}x{ny}x{nt}  {eng.kernel_variant()}; upload-time passes (image + raw pixel sums, once per file): {t_up*1e3:.2f} ms")
print(f"{'chain position':34s} {'want_means':>10s} {'ms / recompute':>15s} {'M traces/s':>11s}")
for means in (1, 0, 2):
    cfg.want_means = means
    sess.recompute(cfg, 1)
    for pos in (1, 3, 5, 6, 7):
        if means == 2 and pos != 1:
            continue
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); sess.recompute(cfg, pos); ts.append(time.perf_counter() - t0)
        dt = sorted(ts)[3]
        print(f"{pos} {names[pos]:32s} {means:10d} {dt*1e3:15.3f} {nx*ny/dt/1e6:11.1f}", flush=True)
sess.close(); eng.close()
