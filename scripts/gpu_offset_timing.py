"""Developer tool: does staggering the base addresses of the five streams of the fused chain change its rate?"""
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
PAD = 8 << 20
d_t = eng.to_device(tm); d_raw = eng.empty((npix, nt)); eng.synth_cube(d_raw, npix, 0, d_t)
d_pre = eng.to_device(chain["w_pre"]); d_fd = eng.to_device(chain["fd_mask"]); d_post = eng.to_device(chain["w_post"])
b_fft = eng.alloc(npix * nf * 8 + PAD); b_amp = eng.alloc(npix * nf * 4 + PAD); b_ph = eng.alloc(npix * nf * 4 + PAD)
b_out = eng.alloc(npix * nt * 4 + PAD); d_img = eng.empty((npix,))
print("bases", [hex(b.ptr) for b in (d_raw, b_fft, b_amp, b_ph, b_out)])
eng.enable_timing(1)
cfgs = {"aligned": (0, 0, 0, 0), "256B steps": (256, 512, 768, 1024), "4K+256 steps": (4352, 8704, 13056, 17408),
        "64K+4K+256": (69888, 139776, 209664, 279552), "1M+64K+256": (1114368, 2228736, 3343104, 4457472), "aligned again": (0, 0, 0, 0)}
for rnd in range(2):
    for name, (o1, o2, o3, o4) in cfgs.items():
        ts = []
        for _ in range(4):
            eng.pipeline(npix, d_raw, d_pre, d_fd, d_post, b_fft.ptr + o1, b_amp.ptr + o2, b_ph.ptr + o3, b_out.ptr + o4, d_img)
            ts.append(eng.stage_time_ns(pkg.binding.STAGE_PIPELINE))
        ms = min(ts) / 1e6
        print(f"{name:16s} {ms:7.3f} ms  {npix * (16 * nt + 20) / ms / 8e7:5.1f} %", flush=True)
eng.close()
