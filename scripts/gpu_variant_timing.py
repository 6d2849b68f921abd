"""Developer tool: the fused chain's variants timed in ONE process on the same buffers (box-to-box differences of
10-15 % otherwise swamp kernel changes): plain / in-launch sums / complex multiplier / both, each with the
store-phase barrier modes of FArgs::bar (THZ_F_BAR), interleaved over several rounds.
Usage: scripts/gpu_variant_timing.py [nx ny nt] ; THZ_VT_BARS="0 1 3" THZ_VT_ROUNDS=5"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from thz_image_explorer_amd import binding, Engine
import synth

nx, ny, nt = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (512, 1024, 4096)))
bars = [int(b) for b in os.environ.get("THZ_VT_BARS", "0 1 2 3 7").split()]
rounds = int(os.environ.get("THZ_VT_ROUNDS", "5"))
eng = Engine(0)
tm = synth.make_time(nt); eng.set_time_axis(tm); nf = eng.nf
chain = synth.default_chain(tm)
npix = nx * ny
d_t = eng.to_device(tm); d_raw = eng.empty((npix, nt)); eng.synth_cube(d_raw, npix, 0, d_t)
d_pre = eng.to_device(chain["w_pre"]); d_fd = eng.to_device(chain["fd_mask"]); d_post = eng.to_device(chain["w_post"])
H = np.zeros((nf, 2), np.float32); H[:, 0] = 0.7; H[:, 1] = 0.3
d_H = eng.to_device(H)
d_fft = eng.empty((npix, nf, 2)); d_amp = eng.empty((npix, nf)); d_ph = eng.empty((npix, nf)); d_out = eng.empty((npix, nt)); d_img = eng.empty((npix,))
d_sums = eng.empty((2 * nf,))
variants = {
    "plain": lambda: eng.pipeline(npix, d_raw, d_pre, d_fd, d_post, d_fft, d_amp, d_ph, d_out, d_img),
    "sums": lambda: eng.pipeline_ex(npix, d_raw, d_pre, d_fd, None, d_post, d_fft, d_amp, d_ph, d_out, d_img, d_sums),
    "cmask": lambda: eng.pipeline_ex(npix, d_raw, d_pre, d_fd, d_H, d_post, d_fft, d_amp, d_ph, d_out, d_img, None),
    "cmask+sums": lambda: eng.pipeline_ex(npix, d_raw, d_pre, d_fd, d_H, d_post, d_fft, d_amp, d_ph, d_out, d_img, d_sums),
}
only = os.environ.get("THZ_VT_ONLY")
if only:
    variants = {k: v for k, v in variants.items() if k in only.split(",")}
res = {}
eng.enable_timing(2)
for r in range(rounds + 1):
    for name, fn in variants.items():
        for bar in bars:
            os.environ["THZ_F_BAR"] = str(bar)
            for _ in range(3):
                fn()
            eng.sync()
            ns, calls = eng.timing_collect(binding.STAGE_PIPELINE)
            eng.timing_collect(binding.STAGE_MEAN)
            if r:
                res.setdefault((name, bar), []).append(ns / calls * 1e-6)
m_full = 16 * nt + 20
print(f"{nx}x{ny}x{nt}  {eng.kernel_variant()}  M_full = {m_full} B/trace; kernel time only (hipEvents around the launch)")
for (name, bar), v in res.items():
    v = np.array(v); med = float(np.median(v))
    print(f"{name:11s} bar={bar}  median {med:7.3f} ms  min {v.min():7.3f}  {npix * m_full / med / 1e6:7.1f} GB/s  frac {npix * m_full / med / 1e6 / 8000:.4f}", flush=True)
eng.close()
