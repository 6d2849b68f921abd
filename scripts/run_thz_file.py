"""Opens a dotTHz file (.thz / .thzimg) the way the reference's OpenFile does (io.rs:496-631 through
libthzio.so), uploads the cube to a session, runs the default chain once and prints what the GUI would
read back: image statistics, the selected pixel's traces, timings.

    python scripts/run_thz_file.py scan.thzimg [px py]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import thz_image_explorer_amd as pkg
from thz_image_explorer_amd import io_binding as tio


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    path = sys.argv[1]
    t0 = time.perf_counter()
    with tio.ScanFile(path) as f:
        tm, g = f.time(), f.geometry()
        cube = f.cube()
        print(f"{path}: group '{f.group_name}' ({f.group_count} in file), {'single pulse' if f.kind else 'scan'} "
              f"{f.nx} x {f.ny} x {f.nt}, dx = {g.dx if g.has_dx else None}, dy = {g.dy if g.has_dy else None}")
    t_read = time.perf_counter() - t0
    px, py = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (cube.shape[0] // 2, cube.shape[1] // 2)
    with pkg.Engine(0) as eng:
        sess = pkg.Session(eng, cube.shape[0], cube.shape[1], tm, g.dx if g.has_dx else 1.0, g.dy if g.has_dy else 1.0)
        t0 = time.perf_counter()
        sess.upload(cube, subtract_bias=True)
        t_up = time.perf_counter() - t0
        cfg = pkg.chain_cfg_default(tm)
        sess.recompute(cfg)  # warm (plans, first launch)
        t0 = time.perf_counter()
        sess.recompute(cfg)
        t_rc = time.perf_counter() - t0
        img = sess.download(pkg.BUF_IMG).reshape(cube.shape[:2])
        pl = sess.plot(px, py)
        print(f"kernels: {eng.kernel_variant()}")
        print(f"read {t_read * 1e3:.1f} ms, upload + bias + image {t_up * 1e3:.1f} ms, recompute {t_rc * 1e3:.2f} ms "
              f"({cube.shape[0] * cube.shape[1] / t_rc / 1e6:.2f} M traces/s)")
        print(f"image: min {img.min():.4g} max {img.max():.4g}; pixel ({px}, {py}): filtered peak "
              f"{np.abs(pl['filtered_signal']).max():.4g} at {tm[np.abs(pl['filtered_signal']).argmax()]:.2f} ps, "
              f"spectrum peak at {np.argmax(pl['filtered_signal_fft']) / (tm[-1] - tm[0]):.3f} THz")
        sess.close()


if __name__ == "__main__":
    main()
