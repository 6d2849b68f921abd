"""Quick on-GPU numerics + timing probe (developer tool, not a test)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from thz_image_explorer_amd import Engine

def unwrap_ref(ph):
    out = np.empty_like(ph); out[0] = ph[0]
    pi = np.float32(np.pi)
    for i in range(1, len(ph)):
        d = np.float32(ph[i] - ph[i-1])
        if d > pi: d -= np.float32(2) * pi
        elif d < -pi: d += np.float32(2) * pi
        out[i] = np.float32(out[i-1] + d)
    return out

eng = Engine(0)
rng = np.random.default_rng(1)
for nt in [4, 8, 64, 128, 256, 1024, 2048, 4096, 8192, 1001, 1000, 97, 3000]:
    npix = 300
    time_ax = (1000 + 0.05 * np.arange(nt)).astype(np.float32)
    eng.set_time_axis(time_ax)
    nf = eng.nf
    x = rng.standard_normal((npix, nt)).astype(np.float32)
    wa = rng.uniform(0.5, 1, nt).astype(np.float32)
    mask = rng.uniform(0, 1, nf).astype(np.float32)
    post = rng.uniform(0.5, 1, nt).astype(np.float32)
    d_x = eng.to_device(x); d_wa = eng.to_device(wa); d_mask = eng.to_device(mask); d_post = eng.to_device(post)
    d_dout = eng.empty((npix, nt)); d_fft = eng.empty((npix, nf, 2)); d_amp = eng.empty((npix, nf)); d_ph = eng.empty((npix, nf))
    eng.fft(npix, d_x, d_wa, None, d_dout, d_fft, d_amp, d_ph, d_mask)
    fft = d_fft.download((npix, nf, 2), np.float32); amp = d_amp.download((npix, nf), np.float32); ph = d_ph.download((npix, nf), np.float32)
    xw = x * wa
    ref = np.fft.rfft(xw.astype(np.float64), axis=1)
    X = fft[..., 0] + 1j * fft[..., 1]
    e_fft = np.abs(X - ref * mask).max() / np.abs(ref).max()
    e_amp = np.abs(amp - np.abs(ref) * mask).max() / np.abs(ref).max()
    e_ph = 0
    for p in range(0, npix, 37):
        r = unwrap_ref(np.angle(ref[p]).astype(np.float32))
        d = ph[p] - r
        d = d - 2 * np.pi * np.round(d / (2 * np.pi))
        e_ph = max(e_ph, np.abs(d).max())
    d_out = eng.empty((npix, nt)); d_img = eng.empty((npix,))
    eng.ifft(npix, d_fft, d_post, d_out, d_img)
    out = d_out.download((npix, nt), np.float32); img = d_img.download((npix,), np.float32)
    ref_t = np.fft.irfft(ref * mask, n=nt, axis=1) * post
    e_inv = np.abs(out - ref_t).max() / np.abs(ref_t).max()
    e_img = np.abs(img - (ref_t ** 2).sum(1)).max() / img.max()
    # fused
    d_out2 = eng.empty((npix, nt)); d_img2 = eng.empty((npix,)); d_fft2 = eng.empty((npix, nf, 2)); d_amp2 = eng.empty((npix, nf)); d_ph2 = eng.empty((npix, nf))
    eng.pipeline(npix, d_x, d_wa, d_mask, d_post, d_fft2, d_amp2, d_ph2, d_out2, d_img2)
    out2 = d_out2.download((npix, nt), np.float32)
    e_pipe = np.abs(out2 - ref_t).max() / np.abs(ref_t).max()
    same = np.array_equal(d_fft2.download((npix, nf, 2), np.float32), fft)
    print(f"nt={nt:5d} {eng.kernel_variant():32s} fft {e_fft:.2e} amp {e_amp:.2e} ph {e_ph:.2e} inv {e_inv:.2e} img {e_img:.2e} pipe {e_pipe:.2e} fft_same={same}", flush=True)
    for b in list(eng._bufs): b.free()
    eng._bufs = []

# timing
import sys as _sys
for fam in (0, 1):
  eng.set_kernel_family(fam)
  for (nx, ny, nt) in [(256, 256, 1024), (1024, 256, 1024), (512, 512, 2048), (512, 512, 4096)]:
      npix = nx * ny
      time_ax = (1000 + 0.05 * np.arange(nt)).astype(np.float32)
      eng.set_time_axis(time_ax); nf = eng.nf
      x = rng.standard_normal((npix, nt)).astype(np.float32)
      d_x = eng.to_device(x)
      w = eng.to_device(np.ones(nt, np.float32)); m = eng.to_device(np.ones(nf, np.float32))
      d_fft = eng.empty((npix, nf, 2)); d_amp = eng.empty((npix, nf)); d_ph = eng.empty((npix, nf)); d_out = eng.empty((npix, nt)); d_img = eng.empty((npix,))
      for name, fn, bytes_per in [
          ("fwd(M_fwd)", lambda: eng.fft(npix, d_x, w, None, None, d_fft, None, None, m), 8 * nt + 8),
          ("fwd(all)", lambda: eng.fft(npix, d_x, w, None, None, d_fft, d_amp, d_ph, m), 4 * nt + 16 * nf),
          ("inv", lambda: eng.ifft(npix, d_fft, w, d_out, d_img), 8 * nf + 4 * nt + 4),
          ("pipeline(M_full)", lambda: eng.pipeline(npix, d_x, w, m, w, d_fft, d_amp, d_ph, d_out, d_img), 16 * nt + 20),
      ]:
          fn(); eng.sync()
          t0 = time.perf_counter()
          reps = 5
          for _ in range(reps): fn()
          eng.sync()
          dt = (time.perf_counter() - t0) / reps
          print(f"fam={fam} {eng.kernel_variant():28s} {nx}x{ny}x{nt} {name:18s} {dt*1e3:8.3f} ms  {npix/dt/1e6:8.2f} Mtraces/s  {npix*bytes_per/dt/1e9:8.1f} GB/s ({npix*bytes_per/dt/8e12*100:.1f}% of 8TB/s)", flush=True)
      for b in list(eng._bufs): b.free()
      eng._bufs = []
eng.close()
