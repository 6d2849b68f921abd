"""Developer check: f32 pixel sums of the fused chain (in-launch accumulation vs the two passes) against float64 sums
of the same stored arrays."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from thz_image_explorer_amd import Engine
import synth
for nx, ny, nt in ((128, 128, 1001), (256, 256, 1024), (256, 256, 4096)):
    time, cube = synth.make_cube(nx, ny, nt)
    npix = nx * ny
    for mode in ("in launch", "two passes"):
        if mode == "two passes":
            os.environ["THZ_NO_FUSED_SUMS"] = "1"
        else:
            os.environ.pop("THZ_NO_FUSED_SUMS", None)
        eng = Engine(0); eng.set_time_axis(time); nf = eng.nf
        chain = synth.default_chain(time)
        d_raw = eng.to_device(cube); d_pre = eng.to_device(chain["w_pre"]); d_fd = eng.to_device(chain["fd_mask"]); d_post = eng.to_device(chain["w_post"])
        d_fft = eng.empty((npix, nf, 2)); d_amp = eng.empty((npix, nf)); d_ph = eng.empty((npix, nf)); d_out = eng.empty((npix, nt)); d_img = eng.empty((npix,))
        d_sums = eng.empty((2 * nf,))
        eng.pipeline_ex(npix, d_raw, d_pre, d_fd, None, d_post, d_fft, d_amp, d_ph, d_out, d_img, d_sums); eng.sync()
        s = d_sums.download((2 * nf,), np.float32).astype(np.float64)
        amp = d_amp.download((npix, nf), np.float32).astype(np.float64); ph = d_ph.download((npix, nf), np.float32).astype(np.float64)
        sa, sp = amp.sum(0), ph.sum(0)
        ea = np.abs(s[:nf] - sa) / np.abs(sa).max(); ep = np.abs(s[nf:] - sp) / np.abs(sp).max()
        # the reference's own order in f32: sum over x then over y
        ra = amp.astype(np.float32).reshape(nx, ny, nf)
        seq = np.zeros((ny, nf), np.float32)
        for x in range(nx):
            seq = seq + ra[x]
        tot = np.zeros(nf, np.float32)
        for y in range(ny):
            tot = tot + seq[y]
        er = np.abs(tot.astype(np.float64) - sa) / np.abs(sa).max()
        print(f"{nx}x{ny}x{nt} {mode:10s}: amplitude sums max rel err {ea.max():.2e} (rms {np.sqrt((ea**2).mean()):.2e}), phase sums {ep.max():.2e} (rms {np.sqrt((ep**2).mean()):.2e});"
              f" reference-order f32 amplitude sums {er.max():.2e}", flush=True)
        eng.close()
