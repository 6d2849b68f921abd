"""Developer tool: the fused chain with its pixel sums — as two passes behind the launch, or inside it (FSums:
ticket-ordered accumulation in LDS) — timed in ONE process on the same buffers, interleaved.
Wall time of pipeline_ex + sync (what a recompute pays), median of several rounds."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from thz_image_explorer_amd import Engine
import synth
nx, ny, nt = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (512, 1024, 4096)))
eng = Engine(0)
tm = synth.make_time(nt); eng.set_time_axis(tm); nf = eng.nf
chain = synth.default_chain(tm)
npix = nx * ny
d_t = eng.to_device(tm); d_raw = eng.empty((npix, nt)); eng.synth_cube(d_raw, npix, 0, d_t)
d_pre = eng.to_device(chain["w_pre"]); d_fd = eng.to_device(chain["fd_mask"]); d_post = eng.to_device(chain["w_post"])
d_fft = eng.empty((npix, nf, 2)); d_amp = eng.empty((npix, nf)); d_ph = eng.empty((npix, nf)); d_out = eng.empty((npix, nt)); d_img = eng.empty((npix,))
d_sums = eng.empty((2 * nf,))
variants = [("no sums, 8 waves", {"NOSUMS": "1"}),
            ("no sums, 7 waves", {"NOSUMS": "1", "THZ_F_BLOCK": "448"}),
            ("sums as two passes", {"THZ_NO_FUSED_SUMS": "1"}),
            ("sums in the launch (7 waves at nt 4096)", {}),
            ("sums in the launch, 6 waves", {"THZ_F_BLOCK": "384"})]
res = {}
keys = ("THZ_F_BLOCK", "THZ_NO_FUSED_SUMS")
for r in range(6):
    for name, env in variants:
        for k in keys:
            os.environ.pop(k, None)
        for k, v in env.items():
            if k != "NOSUMS":
                os.environ[k] = v
        sums = None if "NOSUMS" in env else d_sums
        fn = lambda: eng.pipeline_ex(npix, d_raw, d_pre, d_fd, None, d_post, d_fft, d_amp, d_ph, d_out, d_img, sums)
        fn(); eng.sync()
        t0 = time.perf_counter(); fn(); fn(); eng.sync(); dt = (time.perf_counter() - t0) / 2
        if r:
            res.setdefault(name, []).append(dt * 1e3)
print(f"{nx}x{ny}x{nt}: wall time of thz_pipeline_ex (+ sums) per call, median of {len(next(iter(res.values())))} rounds")
for name, v in res.items():
    print(f"  {name:42s} {np.median(v):8.3f} ms   (min {min(v):.3f})", flush=True)
eng.close()
