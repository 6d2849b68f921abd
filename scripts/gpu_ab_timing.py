"""Developer tool: interleaved A/B timing of the fused chain against the no-arithmetic traffic
probe of the same access shape (thz_traffic_probe), so that clock / thermal drift hits both."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from thz_image_explorer_amd import Engine
import thz_image_explorer_amd as pkg
import synth
nx, ny, nt = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (512, 1024, 4096)))
rounds = int(os.environ.get("THZ_AB_ROUNDS", "6"))
eng = Engine(0)
tm = synth.make_time(nt); eng.set_time_axis(tm); nf = eng.nf
chain = synth.default_chain(tm)
npix = nx * ny
d_t = eng.to_device(tm); d_raw = eng.empty((npix, nt)); eng.synth_cube(d_raw, npix, 0, d_t)
d_pre = eng.to_device(chain["w_pre"]); d_fd = eng.to_device(chain["fd_mask"]); d_post = eng.to_device(chain["w_post"])
nfp = nf + 64  # room for the pitched-row probe variants
d_fft = eng.empty((npix, nfp, 2)); d_amp = eng.empty((npix, nfp)); d_ph = eng.empty((npix, nfp)); d_out = eng.empty((npix, nt)); d_img = eng.empty((npix,))
eng.enable_timing(1)
def probe_cfg(blocks, threads, nt_store, pitch=0):
    def run():
        os.environ["THZ_PROBE_BLOCKS"], os.environ["THZ_PROBE_THREADS"], os.environ["THZ_PROBE_NT"] = str(blocks), str(threads), str(nt_store)
        os.environ["THZ_PROBE_PITCH"] = str(pitch)
        eng.traffic_probe(npix, nt, d_raw, d_fft, d_amp, d_ph, d_out)
    return run


cases = {"pipeline": (lambda: eng.pipeline(npix, d_raw, d_pre, d_fd, d_post, d_fft, d_amp, d_ph, d_out, d_img), pkg.binding.STAGE_PIPELINE),
         "probe": (lambda: eng.traffic_probe(npix, nt, d_raw, d_fft, d_amp, d_ph, d_out), pkg.binding.STAGE_PROBE)}
if os.environ.get("THZ_AB_SWEEP"):
    for b_, t_, n_ in [(256, 512, 0), (256, 512, 1), (512, 512, 0), (1024, 512, 0), (2048, 256, 0), (4096, 256, 0), (8192, 64, 0), (512, 256, 0)]:
        cases[f"p{b_}x{t_}{'nt' if n_ else ''}"] = (probe_cfg(b_, t_, n_), pkg.binding.STAGE_PROBE)
    for pitch in (nf + 3, nf + 31, nf + 63):   # 16-byte, 128-byte and 256-byte aligned rows
        cases[f"pitch{pitch}"] = (probe_cfg(256, 512, 0, pitch), pkg.binding.STAGE_PROBE)
    cases["p256x512 "] = (probe_cfg(256, 512, 0), pkg.binding.STAGE_PROBE)
    del cases["probe"]
acc = {k: [] for k in cases}
for r in range(rounds):
    for name, (fn, stage) in cases.items():
        ts = []
        for _ in range(4):
            fn(); ts.append(eng.stage_time_ns(stage))
        acc[name].append(min(ts) / 1e6)
b = 16 * nt + 20
for name, v in acc.items():
    med = float(np.median(v))
    print(f"{name:12s} per-round min ms: {' '.join(f'{x:6.3f}' for x in v)}  median {med:6.3f} ms  {npix*b/med/1e6:7.1f} GB/s  {npix*b/med/8e7:5.1f} %", flush=True)
eng.close()
