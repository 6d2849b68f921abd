"""Synthetic THz cubes and the default filter chain.

Counter-based (Philox4x32-10, seed 0x7A3D2026) so that any tile can be
generated independently: per-trace parameters come from counter
(trace, 0, 0, 1), noise sample g = trace*nt + i from counter (g>>2, 0, 0, 0),
word g&3.  The same generator exists on the device (thz_synth_cube) for the
full-size bench cube.
"""
import numpy as np

SEED = 0x7A3D2026
M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32(c0, c1, c2, c3, k0=SEED, k1=0):
    """Vectorised Philox4x32-10.  Inputs: uint64 arrays holding 32-bit values."""
    c0 = np.asarray(c0, np.uint64); c1 = np.asarray(c1, np.uint64)
    c2 = np.asarray(c2, np.uint64); c3 = np.asarray(c3, np.uint64)
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for r in range(10):
        if r > 0:
            k0 = (k0 + W0) & 0xFFFFFFFF
            k1 = (k1 + W1) & 0xFFFFFFFF
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)), lo1, (hi0 ^ c3 ^ np.uint64(k1)), lo0
    return c0, c1, c2, c3


def _u01(r):
    """24-bit uniform in (0,1), exact in f32"""
    return ((r >> np.uint64(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)


def make_time(nt, t0=1000.0, dt=0.05):
    return (np.float32(t0) + np.float32(dt) * np.arange(nt, dtype=np.float32)).astype(np.float32)


def make_traces(trace_ids, nt, subtract_bias=True):
    """(len(trace_ids), nt) f32 synthetic traces for global trace indices."""
    ids = np.asarray(trace_ids, np.uint64)
    n = ids.size
    t = make_time(nt)
    r0, r1, r2, _ = philox4x32(ids & MASK32, ids >> np.uint64(32), 0, 1)
    u1, u2, u3 = _u01(r0), _u01(r1), _u01(r2)
    A = (np.float32(1.0) + np.float32(0.5) * u1)[:, None]
    tc = (t[0] + np.float32(10.0) + np.float32(2.0) * u2)[:, None]
    delta = (np.float32(3.0) + np.float32(17.0) * u3)[:, None]
    tau = np.float32(0.35)

    def pulse(tt):
        z = (tt / tau).astype(np.float32)
        return (-z * np.exp(-(z * z), dtype=np.float32)).astype(np.float32)

    tt = t[None, :]
    s = A * pulse(tt - tc) + np.float32(0.3) * A * pulse(tt - tc - delta)
    g = ids[:, None] * np.uint64(nt) + np.arange(nt, dtype=np.uint64)[None, :]
    blk = g >> np.uint64(2)
    w = philox4x32(blk & MASK32, blk >> np.uint64(32), 0, 0)
    lane = (g & np.uint64(3)).astype(np.int64)
    ra = np.where(lane < 2, w[0], w[2])
    rb = np.where(lane < 2, w[1], w[3])
    ua, ub = _u01(ra), _u01(rb)
    rad = np.sqrt(np.float32(-2.0) * np.log(ua, dtype=np.float32), dtype=np.float32)
    ang = np.float32(2.0 * np.pi) * ub
    nrm = np.where((lane & 1) == 0, rad * np.cos(ang, dtype=np.float32), rad * np.sin(ang, dtype=np.float32))
    out = (s + np.float32(0.01) * A * nrm.astype(np.float32)).astype(np.float32)
    if subtract_bias:
        out = out - out[:, :1]  # io.rs:578-586
    return np.ascontiguousarray(out, np.float32)


def make_cube(nx, ny, nt, x0=0, ny_total=None):
    """-> (time, cube (nx,ny,nt)); x0/ny_total place the tile inside a larger grid."""
    ny_total = ny if ny_total is None else ny_total
    ids = ((x0 + np.arange(nx, dtype=np.uint64))[:, None] * np.uint64(ny_total)
           + np.arange(ny, dtype=np.uint64)[None, :]).ravel()
    return make_time(nt), make_traces(ids, nt).reshape(nx, ny, nt)


# --------------------------------------------------------------------------
# default chain (main.rs:194-247 order; parameters of SURVEY §8a'-1)
# --------------------------------------------------------------------------
def default_chain(time, backend=None):
    """Multiplier vectors of the reference's default chain for `time`.
    backend: module with host_* functions (the product's binding by default);
    tests pass the oracle to cross-check the two."""
    if backend is None:
        import thz_image_explorer_amd as backend
    time = np.ascontiguousarray(time, np.float32)
    freq = backend.host_frequency_axis(time)
    w_tilt = backend.host_adapted_blackman(time, 0.0, 7.0)          # tilt_compensation.rs:186-188
    w_tdb = backend.host_td_bandpass(time, float(time[0]), float(time[-1]), 2.0)[0]   # reset(): full range
    w_fft = backend.host_fft_window(time, 0, 1.0, 7.0)             # config.rs:205
    fd = backend.host_fd_bandpass(freq, 0.2, 5.0, 0.1)[0]          # band_pass_fd.rs:51-57
    w_post = backend.host_td_bandpass(time, float(time[0]), float(time[-1]), 0.1)[0]
    pre = ((w_tilt * w_tdb).astype(np.float32) * w_fft).astype(np.float32)
    return dict(time=time, frequency=freq, w_tilt=w_tilt, w_td_before=w_tdb, w_fft=w_fft,
                window_type=0, fft_window=(1.0, 7.0), fd_mask=fd, w_post=w_post, w_pre=pre)


class OracleBackend:
    """default_chain backend built from the oracle's own functions (oracle/thz_oracle.c): the reference side
    of a parity test gets its multiplier vectors from here, the product side from the product's host_*
    functions — neither side's inputs depend on the other"""

    @staticmethod
    def host_frequency_axis(time):
        import oracle_binding as ob
        return ob.frequency_axis(time)

    @staticmethod
    def host_adapted_blackman(axis, lo, hi):
        import oracle_binding as ob
        return ob.apply_adapted_blackman(np.ones(len(axis), np.float32), axis, lo, hi)

    @staticmethod
    def host_td_bandpass(time, low, high, width):
        import oracle_binding as ob
        return ob.td_bandpass_window(time, low, high, width)

    @staticmethod
    def host_fft_window(time, wtype, lo, hi):
        import oracle_binding as ob
        return ob.apply_window(wtype, np.ones(len(time), np.float32), time, lo, hi)

    @staticmethod
    def host_fd_bandpass(freq, low, high, width):
        import oracle_binding as ob
        return ob.fd_bandpass_window(freq, low, high, width)


def oracle_chain(time):
    """the default chain's multiplier vectors computed by the oracle"""
    return default_chain(time, backend=OracleBackend)


def run_gpu_pipeline(eng, cube, chain, want=("fft", "amplitudes", "phases", "data", "img")):
    """Fused chain through the C ABI (thz_pipeline); returns host arrays."""
    nx, ny, nt = cube.shape
    npix = nx * ny
    nf = nt // 2 + 1
    d_raw = eng.to_device(cube)
    d_pre = eng.to_device(chain["w_pre"])
    d_fd = eng.to_device(chain["fd_mask"])
    d_post = eng.to_device(chain["w_post"])
    d_fft = eng.empty((npix, nf, 2)); d_amp = eng.empty((npix, nf)); d_ph = eng.empty((npix, nf))
    d_out = eng.empty((npix, nt)); d_img = eng.empty((npix,))
    eng.pipeline(npix, d_raw, d_pre, d_fd, d_post, d_fft, d_amp, d_ph, d_out, d_img)
    res = dict(fft=d_fft.download((nx, ny, nf, 2), np.float32),
               amplitudes=d_amp.download((nx, ny, nf), np.float32),
               phases=d_ph.download((nx, ny, nf), np.float32),
               data=d_out.download((nx, ny, nt), np.float32),
               img=d_img.download((nx, ny), np.float32),
               variant=eng.kernel_variant())
    for b in (d_raw, d_pre, d_fd, d_post, d_fft, d_amp, d_ph, d_out, d_img):
        b.free()
    return res
