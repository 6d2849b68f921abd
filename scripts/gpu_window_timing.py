"""Developer tool: fused chain with edge-taper windows (fast loop) vs full-length windows (general loop)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import thz_image_explorer_amd as pkg
import synth
nx, ny, nt = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (512, 1024, 4096)))
eng = pkg.Engine(0)
tm = synth.make_time(nt); eng.set_time_axis(tm); nf = eng.nf
chain = synth.default_chain(tm)
npix = nx * ny
d_t = eng.to_device(tm); d_raw = eng.empty((npix, nt)); eng.synth_cube(d_raw, npix, 0, d_t)
d_fd = eng.to_device(chain["fd_mask"])
d_fft = eng.empty((npix, nf, 2)); d_amp = eng.empty((npix, nf)); d_ph = eng.empty((npix, nf)); d_out = eng.empty((npix, nt)); d_img = eng.empty((npix,))
hann = pkg.host_fft_window(tm, 2, 0.0, 0.0)
cases = {"edge pre / edge post": (chain["w_pre"], chain["w_post"]),
         "full pre (Hanning) / edge post": ((chain["w_pre"] * hann).astype(np.float32), chain["w_post"]),
         "edge pre / full post": (chain["w_pre"], hann),
         "no windows": (None, None)}
eng.enable_timing(1)
for name, (pre, post) in cases.items():
    d_pre = eng.to_device(pre) if pre is not None else None
    d_post = eng.to_device(post) if post is not None else None
    ts = []
    for _ in range(6):
        eng.pipeline(npix, d_raw, d_pre, d_fd, d_post, d_fft, d_amp, d_ph, d_out, d_img)
        ts.append(eng.stage_time_ns(pkg.binding.STAGE_PIPELINE))
    ms = min(ts[1:]) / 1e6
    print(f"{name:32s} {ms:7.3f} ms  {npix * (16 * nt + 20) / ms / 8e7:5.1f} % of 8 TB/s", flush=True)
eng.close()
