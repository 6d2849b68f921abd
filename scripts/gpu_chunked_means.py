"""Developer experiment: does running the chain in slabs, each followed at once by its amplitude / phase pixel sums,
let the sums read the slab out of the memory-side cache (256 MB) instead of HBM?  Times one whole pass
(fused chain + both sums over every pixel) for several slab counts; 1 slab = what the session does."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from thz_image_explorer_amd import Engine
import synth
nx, ny, nt = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (1024, 1024, 4096)))
eng = Engine(0)
tm = synth.make_time(nt); eng.set_time_axis(tm); nf = eng.nf
chain = synth.default_chain(tm)
npix = nx * ny
d_t = eng.to_device(tm); d_raw = eng.empty((npix, nt)); eng.synth_cube(d_raw, npix, 0, d_t)
d_pre = eng.to_device(chain["w_pre"]); d_fd = eng.to_device(chain["fd_mask"]); d_post = eng.to_device(chain["w_post"])
d_fft = eng.empty((npix, nf, 2)); d_amp = eng.empty((npix, nf)); d_ph = eng.empty((npix, nf)); d_out = eng.empty((npix, nt)); d_img = eng.empty((npix,))
max_slabs = 1024
d_part = eng.empty((max_slabs, 2, nf))


def one_pass(slabs, sums=True):
    per = npix // slabs
    for c in range(slabs):
        p0 = c * per
        eng.pipeline(per, d_raw.ptr + p0 * nt * 4, d_pre, d_fd, d_post, d_fft.ptr + p0 * nf * 8, d_amp.ptr + p0 * nf * 4,
                     d_ph.ptr + p0 * nf * 4, d_out.ptr + p0 * nt * 4, d_img.ptr + p0 * 4)
        if sums:
            eng.pixel_sum(per, nf, 1, d_amp.ptr + p0 * nf * 4, d_part.ptr + (2 * c) * nf * 4)
            eng.pixel_sum(per, nf, 1, d_ph.ptr + p0 * nf * 4, d_part.ptr + (2 * c + 1) * nf * 4)


ref = None
for slabs in (1, 8, 32, 64, 128, 256, 512, 1024):
    for sums in (False, True):
        one_pass(slabs, sums); eng.sync()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); one_pass(slabs, sums); eng.sync(); ts.append(time.perf_counter() - t0)
        dt = sorted(ts)[len(ts) // 2]
        slab_mb = (npix // slabs) * nf * 8 / 2**20
        print(f"{nx}x{ny}x{nt} slabs {slabs:5d} (amp+phase of a slab {slab_mb:8.1f} MiB) sums={int(sums)}: {dt * 1e3:8.3f} ms", flush=True)
    if slabs == 1 or slabs == 64:
        tot = d_part.download((max_slabs, 2, nf), np.float32)[:slabs].astype(np.float64).sum(axis=0)
        if ref is None:
            ref = tot
        else:
            print("   sums agree with the one-slab pass to", float(np.max(np.abs(tot - ref) / (np.abs(ref) + 1e-30))), flush=True)
eng.close()
