"""Developer tool: interleaved timing of thz_pipeline from TWO builds of the library in one process, on
the same device buffers — box-to-box and placement differences (a few %) otherwise swamp kernel changes.
Usage: scripts/gpu_ab_builds.py <other libthzgpu.so> [nx ny nt]; the first build is the package's own."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import thz_image_explorer_amd as pkg
from thz_image_explorer_amd import binding, Engine
import synth

other = sys.argv[1]
nx, ny, nt = (int(a) for a in (sys.argv[2:5] if len(sys.argv) > 4 else (1024, 1024, 4096)))
rounds = int(os.environ.get("THZ_AB_ROUNDS", "8"))
a = Engine(0)
lib_b = C.CDLL(other)
for name, res, args in binding.SYMBOLS:
    if hasattr(lib_b, name):
        fn = getattr(lib_b, name); fn.restype = res; fn.argtypes = args
b = Engine.__new__(Engine)
b.lib, b.ctx, b._bufs = lib_b, binding._P(), []
assert lib_b.thz_create(0, C.byref(b.ctx)) == 0
tm = synth.make_time(nt)
for e in (a, b):
    e.set_time_axis(tm)
nf = a.nf
chain = synth.default_chain(tm)
npix = nx * ny
d_t = a.to_device(tm); d_raw = a.empty((npix, nt)); a.synth_cube(d_raw, npix, 0, d_t)
d_pre = a.to_device(chain["w_pre"]); d_fd = a.to_device(chain["fd_mask"]); d_post = a.to_device(chain["w_post"])
d_fft = a.empty((npix, nf, 2)); d_amp = a.empty((npix, nf)); d_ph = a.empty((npix, nf)); d_out = a.empty((npix, nt)); d_img = a.empty((npix,))
d_sums = a.empty((2 * nf,))
H = np.zeros((nf, 2), np.float32); H[:, 0] = 0.7; H[:, 1] = 0.3
d_H = a.to_device(H)
variants = {
    "plain": lambda e: e.pipeline(npix, d_raw, d_pre, d_fd, d_post, d_fft, d_amp, d_ph, d_out, d_img),
    "sums": lambda e: e.pipeline_ex(npix, d_raw, d_pre, d_fd, None, d_post, d_fft, d_amp, d_ph, d_out, d_img, d_sums),
    "cmask": lambda e: e.pipeline_ex(npix, d_raw, d_pre, d_fd, d_H, d_post, d_fft, d_amp, d_ph, d_out, d_img, None),
    "cmask+sums": lambda e: e.pipeline_ex(npix, d_raw, d_pre, d_fd, d_H, d_post, d_fft, d_amp, d_ph, d_out, d_img, d_sums),
}
only = os.environ.get("THZ_AB_ONLY")
if only:
    variants = {k: v for k, v in variants.items() if k in only.split(",")}
res = {}
for e in (a, b):
    e.enable_timing(2)
for r in range(rounds + 1):
    for vname, fn in variants.items():
        for name, e in (("this", a), ("other", b)):
            if os.environ.get("THZ_AB_VERBOSE"):
                print(f"round {r} {vname} {name}", flush=True)
            for _ in range(3):
                fn(e)
            e.sync()
            ns, calls = e.timing_collect(binding.STAGE_PIPELINE)
            if r:
                res.setdefault((vname, name), []).append(ns / calls * 1e-6)
m_full = 16 * nt + 20
print(f"{nx}x{ny}x{nt}: kernel time (hipEvents), 'this' = the package's build, 'other' = {other}")
for (vname, name), v in res.items():
    v = np.array(v)
    print(f"{vname:11s} {name:6s} median {np.median(v):7.3f} ms  min {v.min():7.3f}  {npix * m_full / np.median(v) / 1e6:7.1f} GB/s  frac {npix * m_full / np.median(v) / 1e6 / 8000:.4f}", flush=True)
