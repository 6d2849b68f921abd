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
import thz_image_explorer_amd as _pkg
_, _lo, _hi = _pkg.host_fd_bandpass(_pkg.host_frequency_axis(tm), 0.2, 5.0, 0.1)
BAND = (int(_lo), int(_hi))
m_full, m_fwd, m_inv = 16 * nt + 20, 8 * nt + 8, 8 * nf + 4 * nt + 4
variants = {
    "plain": (lambda: eng.pipeline(npix, d_raw, d_pre, d_fd, d_post, d_fft, d_amp, d_ph, d_out, d_img), binding.STAGE_PIPELINE, m_full),
    "sums": (lambda: eng.pipeline_ex(npix, d_raw, d_pre, d_fd, None, d_post, d_fft, d_amp, d_ph, d_out, d_img, d_sums), binding.STAGE_PIPELINE, m_full),
    "cmask+sums": (lambda: eng.pipeline_ex(npix, d_raw, d_pre, d_fd, d_H, d_post, d_fft, d_amp, d_ph, d_out, d_img, d_sums), binding.STAGE_PIPELINE, m_full),
    "cmask+sums+band": (lambda: eng.pipeline_ex(npix, d_raw, d_pre, d_fd, d_H, d_post, d_fft, d_amp, d_ph, d_out, d_img, d_sums, band=BAND), binding.STAGE_PIPELINE, m_full),
    "cmask": (lambda: eng.pipeline_ex(npix, d_raw, d_pre, d_fd, d_H, d_post, d_fft, d_amp, d_ph, d_out, d_img, None), binding.STAGE_PIPELINE, m_full),
    "fwd": (lambda: eng.fft(npix, d_raw, d_pre, None, None, d_fft, None, None, d_fd), binding.STAGE_FFT, m_fwd),
    "fwd+ap": (lambda: eng.fft(npix, d_raw, d_pre, None, None, d_fft, d_amp, d_ph, d_fd), binding.STAGE_FFT, 4 * nt + 16 * nf),
    "inv": (lambda: eng.ifft(npix, d_fft, d_post, d_out, d_img), binding.STAGE_IFFT, m_inv),
}
only = os.environ.get("THZ_VT_ONLY")
if only:
    variants = {k: v for k, v in variants.items() if k in only.split(",")}
res = {}
eng.enable_timing(2)
for r in range(rounds + 1):
    for name, (fn, stage, _) in variants.items():
        for bar in bars:
            os.environ["THZ_F_BAR"] = str(bar)
            for _ in range(3):
                fn()
            eng.sync()
            ns, calls = eng.timing_collect(stage)
            if r:
                res.setdefault((name, bar), []).append(ns / calls * 1e-6)
print(f"{nx}x{ny}x{nt}  {eng.kernel_variant()}  kernel time only (hipEvents around the launch); bytes = algorithmic bytes of each variant")
for (name, bar), v in res.items():
    v = np.array(v); med = float(np.median(v)); mb = variants[name][2]
    print(f"{name:8s} bar={bar}  median {med:7.3f} ms  min {v.min():7.3f}  {npix * mb / med / 1e6:7.1f} GB/s  frac {npix * mb / med / 1e6 / 8000:.4f}", flush=True)
eng.close()
