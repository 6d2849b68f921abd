"""ctypes binding of oracle/libthz_oracle.so — the CPU restatement of the
reference used as the checker (tests, smoke, bench cpu_baseline only)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libthz_oracle.so")

_P = C.c_void_p
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            subprocess.check_call(["make"], cwd=ORACLE_DIR)
        L = C.CDLL(ORACLE_SO)
        L.thz_oracle_blackman_window.restype = C.c_float
        L.thz_oracle_blackman_window.argtypes = [C.c_float, C.c_float]
        L.thz_oracle_ifft_stage.restype = C.c_long
        L.thz_oracle_set_conv_threads(C.c_int(cpu_share()))
        _lib = L
    return _lib


def cpu_share(cap=16):
    """CPUs this process may really use: affinity mask and cgroup quota (a GPU box hands a 16-CPU share of a
    256-thread host to the job), capped"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, cap))


def _p(a):
    return None if a is None else a.ctypes.data_as(_P)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def blackman_window(n, m):
    return lib().thz_oracle_blackman_window(C.c_float(n), C.c_float(m))


def apply_adapted_blackman(signal, time, lo, hi):
    s = f32(signal).copy()
    t = f32(time)
    lib().thz_oracle_apply_adapted_blackman(_p(s), _p(t), C.c_int(s.size), C.c_float(lo), C.c_float(hi))
    return s


def apply_window(wtype, signal, time, lo=0.0, hi=0.0):
    s = f32(signal).copy()
    t = f32(time)
    lib().thz_oracle_apply_window(C.c_int(wtype), _p(s), _p(t), C.c_int(s.size), C.c_float(lo), C.c_float(hi))
    return s


def numpy_unwrap(x, period=2 * np.float32(np.pi)):
    x = f32(x)
    out = np.empty_like(x)
    lib().thz_oracle_numpy_unwrap(_p(x), C.c_int(x.size), C.c_float(period), _p(out))
    return out


def frequency_axis(time):
    t = f32(time)
    out = np.empty(t.size // 2 + 1, np.float32)
    lib().thz_oracle_frequency_axis(_p(t), C.c_int(t.size), _p(out))
    return out


def rfft_f32(x):
    x = f32(x)
    out = np.empty(2 * (x.size // 2 + 1), np.float32)
    lib().thz_oracle_rfft_f32(_p(x), C.c_int(x.size), _p(out))
    return out[0::2] + 1j * out[1::2]


def irfft_f32(X, n):
    Xi = np.empty(2 * X.size, np.float32)
    Xi[0::2] = X.real
    Xi[1::2] = X.imag
    out = np.empty(n, np.float32)
    lib().thz_oracle_irfft_f32(_p(Xi), C.c_int(n), _p(out))
    return out


def rfft_f64(x):
    x = np.ascontiguousarray(x, np.float64)
    out = np.empty(2 * (x.size // 2 + 1), np.float64)
    lib().thz_oracle_rfft_f64(_p(x), C.c_int(x.size), _p(out))
    return out[0::2] + 1j * out[1::2]


def rdft_direct_f64(x):
    x = np.ascontiguousarray(x, np.float64)
    out = np.empty(2 * (x.size // 2 + 1), np.float64)
    lib().thz_oracle_rdft_direct_f64(_p(x), C.c_int(x.size), _p(out))
    return out[0::2] + 1j * out[1::2]


def fft_stage(data, time, wtype=0, lo=1.0, hi=7.0, nthreads=0):
    """math_tools::fft.  data (nx,ny,nt) -> dict(data=windowed, fft, amplitudes, phases)"""
    d = f32(data).copy()
    nx, ny, nt = d.shape
    nf = nt // 2 + 1
    t = f32(time)
    fft = np.empty((nx, ny, nf, 2), np.float32)
    amp = np.empty((nx, ny, nf), np.float32)
    ph = np.empty((nx, ny, nf), np.float32)
    lib().thz_oracle_fft_stage(_p(d), _p(t), nx, ny, nt, C.c_int(wtype), C.c_float(lo), C.c_float(hi),
                               _p(fft), _p(amp), _p(ph), C.c_int(nthreads))
    return dict(data=d, fft=fft, amplitudes=amp, phases=ph)


def fd_bandpass_window(frequency, low, high, width):
    f = f32(frequency)
    w = np.empty(f.size, np.float32)
    lo_i, up_i = C.c_int(), C.c_int()
    lib().thz_oracle_fd_bandpass_window(_p(f), C.c_int(f.size), C.c_double(low), C.c_double(high),
                                        C.c_double(width), _p(w), C.byref(lo_i), C.byref(up_i))
    return w, lo_i.value, up_i.value


def fd_bandpass(fft, amplitudes, frequency, low, high, width):
    f = f32(fft).copy()
    a = f32(amplitudes).copy()
    fr = f32(frequency)
    npix = a.size // fr.size
    lib().thz_oracle_fd_bandpass(_p(f), _p(a), _p(fr), C.c_size_t(npix), C.c_int(fr.size),
                                 C.c_double(low), C.c_double(high), C.c_double(width))
    return f, a


def td_bandpass_window(time, low, high, width):
    t = f32(time)
    w = np.empty(t.size, np.float32)
    lo, hi = C.c_double(low), C.c_double(high)
    lo_i, up_i = C.c_int(), C.c_int()
    lib().thz_oracle_td_bandpass_window(_p(t), C.c_int(t.size), C.byref(lo), C.byref(hi), C.c_double(width),
                                        _p(w), C.byref(lo_i), C.byref(up_i))
    return w, lo.value, hi.value, lo_i.value, up_i.value


def td_bandpass(data, time, low, high, width):
    d = f32(data).copy()
    t = f32(time)
    lo, hi = C.c_double(low), C.c_double(high)
    lib().thz_oracle_td_bandpass(_p(d), _p(t), C.c_size_t(d.size // t.size), C.c_int(t.size),
                                 C.byref(lo), C.byref(hi), C.c_double(width))
    return d, lo.value, hi.value


def intensity(data):
    d = f32(data)
    nt = d.shape[-1]
    img = np.empty(d.shape[:-1], np.float32)
    lib().thz_oracle_intensity(_p(d), C.c_size_t(d.size // nt), C.c_int(nt), _p(img))
    return img


def subtract_bias(data):
    d = f32(data).copy()
    nt = d.shape[-1]
    lib().thz_oracle_subtract_bias(_p(d), C.c_size_t(d.size // nt), C.c_int(nt))
    return d


def pixel_mean(arr, ncomp=1):
    """arr (nx,ny,len[,2]) -> (len[,2]) in the reference's summation order"""
    a = f32(arr)
    nx, ny, ln = a.shape[0], a.shape[1], a.shape[2]
    out = np.empty(ln * ncomp, np.float32)
    lib().thz_oracle_pixel_mean(_p(a), nx, ny, ln, ncomp, _p(out))
    return out.reshape(ln, 2) if ncomp == 2 else out


def roi_mask(poly_xy, scaling, shape0, shape1):
    poly = np.ascontiguousarray(poly_xy, np.uint64).reshape(-1, 2)
    mask = np.zeros((shape0, shape1), np.uint8)
    panic = C.c_int(0)
    lib().thz_oracle_roi_mask(_p(poly), C.c_int(poly.shape[0]), C.c_uint64(scaling), shape0, shape1,
                              _p(mask), C.byref(panic))
    return mask, bool(panic.value)


def average_polygon_roi(data, poly_xy, scaling=1):
    d = f32(data)
    s0, s1, ln = d.shape
    poly = np.ascontiguousarray(poly_xy, np.uint64).reshape(-1, 2)
    out = np.empty(ln, np.float32)
    panic = C.c_int(0)
    lib().thz_oracle_average_polygon_roi(_p(d), s0, s1, ln, _p(poly), C.c_int(poly.shape[0]),
                                         C.c_uint64(scaling), _p(out), C.byref(panic))
    return out


def ifft_stage(fft, nt, nthreads=0):
    f = f32(fft)
    nx, ny = f.shape[0], f.shape[1]
    out = np.empty((nx, ny, nt), np.float32)
    nerr = lib().thz_oracle_ifft_stage(_p(f), nx, ny, nt, _p(out), C.c_int(nthreads))
    return out, int(nerr)


def polar_irfft(amp, phase, nt, zero_dc_imag=False):
    a, p = f32(amp), f32(phase)
    out = np.empty(nt, np.float32)
    lib().thz_oracle_polar_irfft(_p(a), _p(p), C.c_int(nt), C.c_int(int(zero_dc_imag)), _p(out))
    return out


def scale3d(arr, s, ncomp=1):
    a = f32(arr)
    nx, ny, ln = a.shape[0], a.shape[1], a.shape[2]
    out = np.empty((nx // s, ny // s, ln) + ((2,) if ncomp == 2 else ()), np.float32)
    lib().thz_oracle_scale3d(_p(a), nx, ny, ln, ncomp, s, _p(out))
    return out


def tilt(data, time, tilt_x_deg, tilt_y_deg, dx, dy):
    """TiltCompensation::filter -> (num_steps, new_time, extended cube)"""
    d = f32(data)
    nx, ny, nt = d.shape
    t = f32(time)
    L = lib()
    L.thz_oracle_tilt.restype = C.c_int
    args = [_p(d), _p(t), nx, ny, nt, C.c_double(tilt_x_deg), C.c_double(tilt_y_deg), C.c_float(dx), C.c_float(dy)]
    steps = L.thz_oracle_tilt(*args, None, None)
    new_time = np.empty(nt + 2 * steps, np.float32)
    out = np.empty((nx, ny, nt + 2 * steps), np.float32)
    L.thz_oracle_tilt(*args, _p(new_time), _p(out))
    return steps, new_time, out


# ---- deconvolution (oracle/thz_oracle_deconv.c) --------------------------------
class OSpline(C.Structure):
    _fields_ = [("n", C.c_int), ("knots", C.c_void_p), ("values", C.c_void_p), ("a", C.c_void_p),
                ("b", C.c_void_p), ("c", C.c_void_p), ("d", C.c_void_p)]


class OHybrid(C.Structure):
    _fields_ = [("base_a", C.c_float), ("base_b", C.c_float), ("corr", OSpline)]


class OPsf(C.Structure):
    _fields_ = [("wx", OHybrid), ("wy", OHybrid), ("x0", OSpline), ("y0", OSpline)]


def psf_from_npz(z):
    keep = []

    def arr(key):
        a = np.ascontiguousarray(np.asarray(z[key], np.float64).astype(np.float32))
        keep.append(a)
        return a

    def spline(prefix, kk, vk):
        k, v = arr(kk), arr(vk)
        co = [arr(f"{prefix}coeff_{c}") for c in "abcd"]
        return OSpline(k.size, k.ctypes.data, v.ctypes.data, *[c.ctypes.data for c in co])

    p = OPsf()
    p.wx = OHybrid(float(np.asarray(z["wx_base_a"]).ravel()[0]), float(np.asarray(z["wx_base_b"]).ravel()[0]),
                   spline("wx_corr_", "wx_corr_knots_thz", "wx_corr_values_mm"))
    p.wy = OHybrid(float(np.asarray(z["wy_base_a"]).ravel()[0]), float(np.asarray(z["wy_base_b"]).ravel()[0]),
                   spline("wy_corr_", "wy_corr_knots_thz", "wy_corr_values_mm"))
    p.x0 = spline("x0_", "x0_knots_thz", "x0_values_mm")
    p.y0 = spline("y0_", "y0_knots_thz", "y0_values_mm")
    p._keep = keep
    return p


def psf_eval(psf, freqs):
    f = f32(freqs)
    out = [np.empty(f.size, np.float32) for _ in range(4)]
    lib().thz_oracle_psf_eval(C.byref(psf), _p(f), C.c_int(f.size), *[_p(o) for o in out])
    return tuple(out)


def filter_bank(time, n_filters, start_freq, end_freq, win_width):
    t = f32(time)
    filters = np.empty((n_filters, 499), np.float32)
    centers = np.empty(n_filters, np.float32)
    # the filter's fields are f32 and widened with `as f64` (deconvolution.rs:823-825)
    lib().thz_oracle_filter_bank(C.c_int(n_filters), C.c_double(float(np.float32(start_freq))),
                                 C.c_double(float(np.float32(end_freq))), C.c_double(float(np.float32(win_width))),
                                 _p(t), _p(filters), _p(centers))
    return filters, centers


def band_psf(psf, center_freq, dx, dy, img_rows, img_cols):
    r, c, wx = C.c_int(), C.c_int(), C.c_float()
    L = lib()
    L.thz_oracle_band_psf(C.byref(psf), C.c_float(center_freq), C.c_float(dx), C.c_float(dy), img_rows, img_cols,
                          None, C.byref(r), C.byref(c), C.byref(wx))
    out = np.empty((r.value, c.value), np.float32)
    L.thz_oracle_band_psf(C.byref(psf), C.c_float(center_freq), C.c_float(dx), C.c_float(dy), img_rows, img_cols,
                          _p(out), C.byref(r), C.byref(c), C.byref(wx))
    return out


def filter_scan(data, filt):
    d = f32(data)
    nt = d.shape[-1]
    f = f32(filt)
    out = np.empty_like(d)
    lib().thz_oracle_filter_scan(_p(d), C.c_size_t(d.size // nt), C.c_int(nt), _p(f), C.c_int(f.size), _p(out))
    return out


def richardson_lucy(image, psf2d, n_iter):
    im, ps = f32(image), f32(psf2d)
    out = np.empty_like(im)
    lib().thz_oracle_richardson_lucy(_p(im), im.shape[0], im.shape[1], _p(ps), ps.shape[0], ps.shape[1],
                                     C.c_int(n_iter), _p(out))
    return out


def deconvolution(data, time, dx, dy, psf, n_iterations, n_filters, start_freq, end_freq, win_width):
    """-> (status, out cube, img, gains (n_filters,nx,ny), n_iter per band)"""
    d = f32(data)
    nx, ny, nt = d.shape
    t = f32(time)
    out = np.empty_like(d)
    img = np.empty((nx, ny), np.float32)
    gains = np.zeros((n_filters, nx, ny), np.float32)
    niter = np.zeros(n_filters, np.int32)
    L = lib()
    L.thz_oracle_deconvolution.restype = C.c_int
    rc = L.thz_oracle_deconvolution(_p(d), _p(t), nx, ny, nt, C.c_float(dx), C.c_float(dy), C.byref(psf),
                                    C.c_int(n_iterations), C.c_int(n_filters), C.c_double(float(np.float32(start_freq))),
                                    C.c_double(float(np.float32(end_freq))), C.c_double(float(np.float32(win_width))),
                                    _p(out), _p(img), _p(gains),
                                    _p(niter))
    return rc, out, img, gains, niter


# ---- 3-D voxel envelope (oracle/thz_oracle_voxel.c) ------------------------------
VOXEL_INSTANCE = np.dtype([("position", np.float32, 3), ("scale", np.float32), ("color", np.float32, 4)])


def gaussian_kernel1d(sigma, radius):
    out = np.empty(2 * radius + 1, np.float32)
    lib().thz_oracle_gaussian_kernel1d(C.c_float(sigma), C.c_int(radius), _p(out))
    return out


def voxel_opacity(data, sigma=3.0, radius=9, contrast=2.0, opacity_threshold=0.1):
    d = f32(data)
    nt = d.shape[-1]
    out = np.empty_like(d)
    lib().thz_oracle_voxel_opacity(_p(d), C.c_size_t(d.size // nt), C.c_int(nt), C.c_float(sigma), C.c_int(radius),
                                   C.c_float(contrast), C.c_float(opacity_threshold), _p(out))
    return out


def voxel_threshold(opacity, max_instances=2_000_000):
    o = f32(opacity)
    L = lib()
    L.thz_oracle_voxel_threshold.restype = C.c_float
    return float(L.thz_oracle_voxel_threshold(_p(o), C.c_size_t(o.size), C.c_size_t(max_instances)))


def voxel_instances(opacity, threshold, time_span, scaling, orig_dims):
    """-> (instances structured array, (cube_width, cube_height, cube_depth))"""
    o = f32(opacity)
    gw, gh, gd = o.shape
    L = lib()
    L.thz_oracle_voxel_instances.restype = C.c_size_t
    dims = np.zeros(3, np.float32)
    args = [_p(o), C.c_size_t(gw), C.c_size_t(gh), C.c_size_t(gd), C.c_float(threshold), C.c_float(time_span),
            C.c_int(scaling), C.c_size_t(orig_dims[0]), C.c_size_t(orig_dims[1]), C.c_size_t(orig_dims[2])]
    n = L.thz_oracle_voxel_instances(*args, None, C.c_size_t(0), _p(dims))
    out = np.zeros(n, VOXEL_INSTANCE)
    L.thz_oracle_voxel_instances(*args, _p(out), C.c_size_t(n), _p(dims))
    return out, tuple(float(x) for x in dims)


def open_ref(scan_time, ref_time, ref_signal, wtype=0, lo=1.0, hi=7.0):
    """OpenRef -> (reference, amplitudes, phases, align_mode) or None where the reference panics"""
    st, rt, rs = f32(scan_time), f32(ref_time), f32(ref_signal)
    nt, nf = st.size, st.size // 2 + 1
    ref, amp, ph = np.empty(nt, np.float32), np.empty(nf, np.float32), np.empty(nf, np.float32)
    mode = C.c_int()
    rc = lib().thz_oracle_open_ref(_p(st), C.c_int(nt), _p(rt), _p(rs), C.c_int(rs.size), C.c_int(wtype),
                                   C.c_float(lo), C.c_float(hi), _p(ref), _p(amp), _p(ph), C.byref(mode))
    return None if rc else (ref, amp, ph, mode.value)


def optical_properties(sample_amp, sample_phase, ref_amp, ref_phase, freq, thickness):
    a, p, ra, rp, f = (f32(x) for x in (sample_amp, sample_phase, ref_amp, ref_phase, freq))
    out = [np.empty(f.size, np.float32) for _ in range(3)]
    lib().thz_oracle_optical_properties(_p(a), _p(p), _p(ra), _p(rp), _p(f), C.c_size_t(f.size), C.c_float(thickness),
                                        *[_p(o) for o in out])
    return tuple(out)


def max_threads():
    return int(lib().thz_oracle_max_threads())


def run_pipeline(cube, time, chain, nthreads=0):
    """Default chain, stage-fused per trace (thz_oracle_pipeline).  `chain` is
    the dict built by synth.default_chain()."""
    d = f32(cube)
    nx, ny, nt = d.shape
    nf = nt // 2 + 1
    t = f32(time)
    fft = np.empty((nx, ny, nf, 2), np.float32)
    amp = np.empty((nx, ny, nf), np.float32)
    ph = np.empty((nx, ny, nf), np.float32)
    out = np.empty((nx, ny, nt), np.float32)
    img = np.empty((nx, ny), np.float32)
    tilt = None if chain.get("w_tilt") is None else f32(chain["w_tilt"])
    lib().thz_oracle_pipeline(_p(d), _p(t), nx, ny, nt, _p(tilt), _p(f32(chain["w_td_before"])),
                              C.c_int(chain["window_type"]),
                              C.c_float(chain["fft_window"][0]), C.c_float(chain["fft_window"][1]),
                              _p(f32(chain["fd_mask"])), _p(f32(chain["w_post"])), _p(fft), _p(amp),
                              _p(ph), _p(out), _p(img), C.c_int(nthreads))
    return dict(fft=fft, amplitudes=amp, phases=ph, data=out, img=img)
