"""Developer tool: device time and achieved GB/s of the voxel-envelope kernels (K15)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import synth
import thz_image_explorer_amd as pkg

nx, ny, nt = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (1024, 256, 4096)))
npix = nx * ny
eng = pkg.Engine(0)
t = synth.make_time(nt)
d_time = eng.to_device(t)
d_cube = eng.empty((npix, nt))
eng.set_time_axis(t)
eng.synth_cube(d_cube, npix, 0, d_time)
d_gain = eng.to_device(np.full(nt, 4.0, np.float32))
eng.apply_td_window(npix, d_cube, d_gain, d_cube)
d_op = eng.empty((npix, nt))
cfg = pkg.voxel_cfg_default()
eng.enable_timing(1)
n = npix * nt
for name, fn, stage, bytes_ in [
    ("opacity", lambda: eng.voxel_opacity(npix, nt, d_cube, cfg, d_op), pkg.binding.STAGE_VOXEL_OPACITY, 8 * n),
    ("threshold", lambda: eng.voxel_threshold(d_op, n), pkg.binding.STAGE_VOXEL_SELECT, 12 * n),
]:
    for _ in range(2):
        fn()
    ts = []
    for _ in range(5):
        fn()
        ts.append(eng.stage_time_ns(stage))
    ms = np.median(ts) / 1e6
    print(f"{name:10s} {ms:8.3f} ms  {bytes_ / ms / 1e9:7.2f} TB/s-equivalent ({bytes_ / 1e9:.1f} GB algorithmic)")
thr = eng.voxel_threshold(d_op, n)
cap = 4_000_000
d_inst = eng.alloc(cap * 32)
ts = []
for _ in range(5):
    cnt, _ = eng.voxel_instances(d_op, nx, ny, nt, thr, float(t[-1] - t[0]), 1, (nx, ny, nt), d_inst, cap)
    ts.append(eng.stage_time_ns(pkg.binding.STAGE_VOXEL_EMIT))
ms = np.median(ts) / 1e6
print(f"{'instances':10s} {ms:8.3f} ms  {8 * n / ms / 1e9:7.2f} TB/s-equivalent  count={cnt} thr={thr}")
