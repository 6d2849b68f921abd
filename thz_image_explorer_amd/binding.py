"""ctypes binding of libthzgpu.so (include/thzgpu.h).

Test / bench harness glue only: the product is the C-ABI library.  There is no
CPU fallback — if the library is missing or no GPU is visible, loading or
`Engine()` raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libthzgpu.so")

THZ_OK = 0
STATUS = {0: "THZ_OK", -1: "THZ_ERR_INVALID", -2: "THZ_ERR_UNSUPPORTED", -3: "THZ_ERR_HIP",
          -4: "THZ_ERR_NOT_READY", -5: "THZ_ERR_ABORTED"}

WIN_ADAPTED_BLACKMAN, WIN_BLACKMAN, WIN_HANNING, WIN_HAMMING, WIN_FLAT_TOP = range(5)
STAGE_FFT, STAGE_FD_MASK, STAGE_IFFT, STAGE_PIPELINE, STAGE_TD_WINDOW, STAGE_INTENSITY, \
    STAGE_MEAN, STAGE_ROI, STAGE_VOXEL_OPACITY, STAGE_VOXEL_SELECT, STAGE_VOXEL_EMIT, STAGE_PROBE = range(12)


class WindowCfg(C.Structure):
    _fields_ = [("type", C.c_int32), ("lower", C.c_float), ("upper", C.c_float)]


class Spline(C.Structure):
    _fields_ = [("knots", C.c_void_p), ("values", C.c_void_p), ("coeff_a", C.c_void_p), ("coeff_b", C.c_void_p),
                ("coeff_c", C.c_void_p), ("coeff_d", C.c_void_p), ("n_knots", C.c_size_t)]


class HybridFit(C.Structure):
    _fields_ = [("base_a", C.c_float), ("base_b", C.c_float), ("correction", Spline)]


class Psf(C.Structure):
    _fields_ = [("wx_fit", HybridFit), ("wy_fit", HybridFit), ("x0_spline", Spline), ("y0_spline", Spline)]


class DeconvCfg(C.Structure):
    _fields_ = [("n_iterations", C.c_uint32), ("n_filters", C.c_uint32), ("start_freq", C.c_float),
                ("end_freq", C.c_float), ("win_width", C.c_float), ("band_begin", C.c_uint32),
                ("band_end", C.c_uint32)]


def psf_from_npz(z) -> "Psf":
    """Builds a thz_psf from the arrays of a psf.npz (keys of io.rs:146-166).
    The f32 copies are kept alive on the returned object."""
    keep = []

    def arr(key):
        a = np.ascontiguousarray(np.asarray(z[key], np.float64).astype(np.float32))
        keep.append(a)
        return a

    def spline(prefix, knots_key, values_key):
        k, v = arr(knots_key), arr(values_key)
        co = [arr(f"{prefix}coeff_{c}") for c in "abcd"]
        return Spline(k.ctypes.data, v.ctypes.data, co[0].ctypes.data, co[1].ctypes.data, co[2].ctypes.data,
                      co[3].ctypes.data, k.size)

    psf = Psf()
    psf.wx_fit = HybridFit(float(np.asarray(z["wx_base_a"]).ravel()[0]), float(np.asarray(z["wx_base_b"]).ravel()[0]),
                           spline("wx_corr_", "wx_corr_knots_thz", "wx_corr_values_mm"))
    psf.wy_fit = HybridFit(float(np.asarray(z["wy_base_a"]).ravel()[0]), float(np.asarray(z["wy_base_b"]).ravel()[0]),
                           spline("wy_corr_", "wy_corr_knots_thz", "wy_corr_values_mm"))
    psf.x0_spline = spline("x0_", "x0_knots_thz", "x0_values_mm")
    psf.y0_spline = spline("y0_", "y0_knots_thz", "y0_values_mm")
    psf._keep = keep
    return psf


class VoxelCfg(C.Structure):
    _fields_ = [("opacity_threshold", C.c_float), ("contrast", C.c_float), ("sigma", C.c_float),
                ("radius", C.c_int32)]


VOXEL_INSTANCE = np.dtype([("position", np.float32, 3), ("scale", np.float32), ("color", np.float32, 4)])
VOXEL_MAX_INSTANCES = 2_000_000


class ChainCfg(C.Structure):
    _fields_ = [("tilt_active", C.c_int32), ("tilt_x_deg", C.c_double), ("tilt_y_deg", C.c_double),
                ("td_before_active", C.c_int32), ("td_before_low", C.c_double), ("td_before_high", C.c_double),
                ("td_before_width", C.c_double), ("fft_window", WindowCfg),
                ("fd_active", C.c_int32), ("fd_low", C.c_double), ("fd_high", C.c_double), ("fd_width", C.c_double),
                ("td_after_active", C.c_int32), ("td_after_low", C.c_double), ("td_after_high", C.c_double),
                ("td_after_width", C.c_double), ("want_means", C.c_int32), ("scale_factor", C.c_int32),
                ("avg_in_fourier_space", C.c_int32)]


class PipelineIo(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("d_raw", "d_pre_win", "d_fd_mask", "d_fd_cmask", "d_post_win", "d_fft", "d_amp",
                                          "d_phase", "d_data_out", "d_img", "d_sums")] + [("band_lo", C.c_size_t), ("band_hi", C.c_size_t)]


class RoiOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("signal_fft", "phase_fft", "signal", "roi_data", "count")]


class PlotOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("signal", "signal_fft", "phase_fft", "filtered_signal", "filtered_signal_fft",
                                  "filtered_phase_fft", "avg_signal", "avg_signal_fft", "avg_phase_fft")]


BUF_RAW, BUF_FFT, BUF_AMPLITUDES, BUF_PHASES, BUF_DATA, BUF_IMG, BUF_AVG_FFT, BUF_AVG_AMPLITUDES, \
    BUF_AVG_PHASES, BUF_OPACITY = range(10)


class ThzError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{STATUS.get(code, code)}: {msg}")
        self.code = code


# every symbol include/thzgpu.h declares: (name, restype, argtypes)
_P = C.c_void_p
_SZ = C.c_size_t
SYMBOLS = [
    ("thz_abi_version", C.c_int, []),
    ("thz_create", C.c_int, [C.c_int, C.POINTER(_P)]),
    ("thz_destroy", None, [_P]),
    ("thz_release_scratch", C.c_int, [_P]),
    ("thz_last_error", C.c_char_p, [_P]),
    ("thz_stream", _P, [_P]),
    ("thz_sync", C.c_int, [_P]),
    ("thz_malloc", C.c_int, [_P, C.POINTER(_P), _SZ]),
    ("thz_free", C.c_int, [_P, _P]),
    ("thz_memcpy_h2d", C.c_int, [_P, _P, _P, _SZ]),
    ("thz_memcpy_d2h", C.c_int, [_P, _P, _P, _SZ]),
    ("thz_memcpy_d2d", C.c_int, [_P, _P, _P, _SZ]),
    ("thz_memset", C.c_int, [_P, _P, C.c_int, _SZ]),
    ("thz_set_time_axis", C.c_int, [_P, _P, _SZ]),
    ("thz_set_kernel_family", C.c_int, [_P, C.c_int]),
    ("thz_nt", _SZ, [_P]),
    ("thz_nf", _SZ, [_P]),
    ("thz_get_frequency", C.c_int, [_P, _P]),
    ("thz_host_frequency_axis", C.c_int, [_P, _SZ, _P]),
    ("thz_host_fft_window", C.c_int, [_P, _SZ, C.POINTER(WindowCfg), _P]),
    ("thz_host_adapted_blackman", C.c_int, [_P, _SZ, C.c_float, C.c_float, _P]),
    ("thz_host_td_bandpass", C.c_int, [_P, _SZ, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                       C.c_double, _P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("thz_host_fd_bandpass", C.c_int, [_P, _SZ, C.c_double, C.c_double, C.c_double, _P,
                                       C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("thz_host_water_line_mask", C.c_int, [_P, _SZ, _P, _SZ, C.c_float, _P]),
    ("thz_host_wiener_filter", C.c_int, [_P, _SZ, C.c_float, _P]),
    ("thz_host_tilt_plan", _SZ, [_P, _SZ, _SZ, _SZ, C.c_double, C.c_double, C.c_float, C.c_float, _P, _P]),
    ("thz_tilt_apply", C.c_int, [_P, _SZ, _P, _SZ, _P, _P, _SZ, _P]),
    ("thz_fft", C.c_int, [_P, _SZ, _P, _P, _P, _P, _P, _P, _P, _P]),
    ("thz_apply_fd_mask", C.c_int, [_P, _SZ, _P, _P, _P]),
    ("thz_apply_fd_cmask", C.c_int, [_P, _SZ, _P, _P, _P]),
    ("thz_ifft", C.c_int, [_P, _SZ, _P, _P, _P, _P]),
    ("thz_pipeline", C.c_int, [_P, _SZ, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    ("thz_pipeline_ex", C.c_int, [_P, _SZ, _P]),
    ("thz_apply_td_window", C.c_int, [_P, _SZ, _P, _P, _P]),
    ("thz_intensity", C.c_int, [_P, _SZ, _P, _P]),
    ("thz_subtract_bias", C.c_int, [_P, _SZ, _P, _P]),
    ("thz_pixel_mean", C.c_int, [_P, _SZ, _SZ, _SZ, C.c_int, _P, _P]),
    ("thz_pixel_sum", C.c_int, [_P, _SZ, _SZ, C.c_int, _P, _P]),
    ("thz_roi_mask", C.c_int, [_P, _P, _SZ, C.c_uint64, _SZ, _SZ, _P]),
    ("thz_roi_mean", C.c_int, [_P, _P, _SZ, _SZ, _SZ, _P, _P, _P, C.c_int]),
    ("thz_scale3d", C.c_int, [_P, _P, _SZ, _SZ, _SZ, C.c_int, _SZ, _P]),
    ("thz_host_psf_eval", C.c_int, [C.POINTER(Psf), _P, _SZ, _P, _P, _P, _P]),
    ("thz_host_filter_bank", C.c_int, [_P, _SZ, C.POINTER(DeconvCfg), _P, _P]),
    ("thz_host_band_psf", C.c_int, [C.POINTER(Psf), C.c_float, C.c_float, C.c_float, _SZ, _SZ, _P,
                                    C.POINTER(_SZ), C.POINTER(_SZ)]),
    ("thz_polar_ifft", C.c_int, [_P, _P, _P, C.c_int, _P]),
    ("thz_deconvolve", C.c_int, [_P, C.POINTER(Psf), C.POINTER(DeconvCfg), _SZ, _SZ, C.c_float, C.c_float,
                                 _P, _P, _P, _P, _P, _P]),
    ("thz_synth_cube", C.c_int, [_P, _P, _SZ, C.c_uint64, _P, C.c_uint32, C.c_int]),
    ("thz_chain_cfg_default", C.c_int, [_P, _SZ, C.POINTER(ChainCfg)]),
    ("thz_session_create", C.c_int, [_P, _SZ, _SZ, _SZ, _P, C.c_float, C.c_float, C.POINTER(_P)]),
    ("thz_session_destroy", None, [_P]),
    ("thz_session_upload", C.c_int, [_P, _P, C.c_int]),
    ("thz_session_recompute", C.c_int, [_P, C.POINTER(ChainCfg)]),
    ("thz_session_recompute_from", C.c_int, [_P, C.POINTER(ChainCfg), C.c_int]),
    ("thz_session_set_fd_filters", C.c_int, [_P, _P, _P, _SZ]),
    ("thz_session_set_rois", C.c_int, [_P, _SZ, _P, _P]),
    ("thz_session_roi_count", _SZ, [_P]),
    ("thz_session_roi", C.c_int, [_P, _SZ, C.POINTER(RoiOut)]),
    ("thz_session_grid", C.c_int, [_P, C.POINTER(_SZ), C.POINTER(_SZ), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    ("thz_session_deconvolve", C.c_int, [_P, C.POINTER(Psf), C.POINTER(DeconvCfg), _P, _P]),
    ("thz_session_nt_out", _SZ, [_P]),
    ("thz_session_time_out", C.c_int, [_P, _P]),
    ("thz_session_buffer", _P, [_P, C.c_int]),
    ("thz_session_download", C.c_int, [_P, C.c_int, _SZ, _SZ, _P]),
    ("thz_host_slab", C.c_int, [_SZ, C.c_int, C.c_int, C.POINTER(_SZ), C.POINTER(_SZ)]),
    ("thz_group_create", C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(_P)]),
    ("thz_group_unique_id", C.c_int, [_P]),
    ("thz_group_create_rank", C.c_int, [C.c_int, C.c_int, C.c_int, _P, C.POINTER(_P)]),
    ("thz_group_destroy", None, [_P]),
    ("thz_group_last_error", C.c_char_p, [_P]),
    ("thz_group_world", C.c_int, [_P]),
    ("thz_group_local_count", C.c_int, [_P]),
    ("thz_group_rank", C.c_int, [_P, C.c_int]),
    ("thz_group_ctx", _P, [_P, C.c_int]),
    ("thz_group_all_reduce_sum", C.c_int, [_P, C.POINTER(_P), _SZ]),
    ("thz_group_all_reduce_u64", C.c_int, [_P, C.POINTER(_P), _SZ]),
    ("thz_group_gather", C.c_int, [_P, C.POINTER(_P), C.POINTER(_SZ), _P]),
    ("thz_group_sync", C.c_int, [_P]),
    ("thz_group_session_create", C.c_int, [_P, _SZ, _SZ, _SZ, _P, C.c_float, C.c_float, C.POINTER(_P)]),
    ("thz_group_session_destroy", None, [_P]),
    ("thz_group_session_member", _P, [_P, C.c_int]),
    ("thz_group_session_upload", C.c_int, [_P, _P, C.c_int]),
    ("thz_group_session_recompute", C.c_int, [_P, C.POINTER(ChainCfg), C.c_int, C.c_int]),
    ("thz_group_session_deconvolve", C.c_int, [_P, C.POINTER(Psf), C.POINTER(DeconvCfg), _P, _P]),
    ("thz_group_session_result", _P, [_P, C.c_int]),
    ("thz_group_session_download", C.c_int, [_P, C.c_int, _SZ, _SZ, _P]),
    ("thz_group_session_grid", C.c_int, [_P, C.POINTER(_SZ), C.POINTER(_SZ)]),
    ("thz_group_session_set_rois", C.c_int, [_P, _SZ, _P, _P]),
    ("thz_group_session_roi", C.c_int, [_P, _SZ, C.POINTER(RoiOut)]),
    ("thz_session_plot", C.c_int, [_P, _SZ, _SZ, C.POINTER(PlotOut)]),
    ("thz_session_voxels", C.c_int, [_P, C.POINTER(VoxelCfg), C.c_uint64, C.c_int, _SZ, _SZ, _SZ, _P, C.c_uint64,
                                     C.POINTER(C.c_uint64), C.POINTER(C.c_float), _P]),
    ("thz_host_align_reference", C.c_int, [_P, _SZ, _P, _P, _SZ, _P]),
    ("thz_reference_spectrum", C.c_int, [_P, _P, _SZ, _P, _P, _SZ, C.POINTER(WindowCfg), _P, _P, _P]),
    ("thz_host_optical_properties", C.c_int, [_P, _P, _P, _P, _P, _SZ, C.c_float, _P, _P, _P]),
    ("thz_traffic_probe", C.c_int, [_P, _SZ, _SZ, _P, _P, _P, _P, _P]),
    ("thz_voxel_cfg_default", C.c_int, [C.POINTER(VoxelCfg)]),
    ("thz_host_gaussian_kernel1d", C.c_int, [C.c_float, C.c_int, _P]),
    ("thz_voxel_opacity", C.c_int, [_P, _SZ, _SZ, _P, C.POINTER(VoxelCfg), _P]),
    ("thz_select_histogram", C.c_int, [_P, _P, _SZ, C.c_int, C.c_uint32, _P]),
    ("thz_host_select_step", C.c_int, [_P, C.c_int, C.c_uint64, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
    ("thz_host_select_value", C.c_float, [C.c_int, C.c_int, C.c_int]),
    ("thz_kth_largest", C.c_int, [_P, _P, _SZ, C.c_uint64, C.POINTER(C.c_float)]),
    ("thz_voxel_threshold", C.c_int, [_P, _P, _SZ, C.c_uint64, C.POINTER(C.c_float)]),
    ("thz_voxel_instances", C.c_int, [_P, _P, _SZ, _SZ, _SZ, _SZ, _SZ, C.c_float, C.c_float, C.c_int, _SZ, _SZ,
                                      _SZ, _P, C.c_uint64, C.POINTER(C.c_uint64), _P]),
    ("thz_enable_timing", C.c_int, [_P, C.c_int]),
    ("thz_stage_time_ns", C.c_int, [_P, C.c_int, C.POINTER(C.c_uint64)]),
    ("thz_timing_collect", C.c_int, [_P, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("thz_kernel_variant", C.c_char_p, [_P]),
]

_lib = None


def load_library() -> C.CDLL:
    """Loads libthzgpu.so and declares every prototype.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _rc(rc: int, what: str):
    if rc != THZ_OK:
        raise ThzError(rc, what)


# -- host multipliers (no GPU, no context) ---------------------------------
def host_frequency_axis(time: np.ndarray) -> np.ndarray:
    t = np.ascontiguousarray(time, np.float32)
    out = np.empty(t.size // 2 + 1, np.float32)
    _rc(load_library().thz_host_frequency_axis(t.ctypes.data, t.size, out.ctypes.data), "frequency_axis")
    return out


def host_fft_window(time, wtype=WIN_ADAPTED_BLACKMAN, lower=1.0, upper=7.0) -> np.ndarray:
    t = np.ascontiguousarray(time, np.float32)
    cfg = WindowCfg(int(wtype), float(lower), float(upper))
    out = np.empty(t.size, np.float32)
    _rc(load_library().thz_host_fft_window(t.ctypes.data, t.size, C.byref(cfg), out.ctypes.data), "fft_window")
    return out


def host_adapted_blackman(axis, lower: float, upper: float) -> np.ndarray:
    a = np.ascontiguousarray(axis, np.float32)
    out = np.empty(a.size, np.float32)
    _rc(load_library().thz_host_adapted_blackman(a.ctypes.data, a.size, lower, upper, out.ctypes.data),
        "adapted_blackman")
    return out


def host_td_bandpass(time, low: float, high: float, width: float):
    """-> (multiplier, clamped_low, clamped_high, lower_idx, upper_idx)"""
    t = np.ascontiguousarray(time, np.float32)
    lo, hi = C.c_double(low), C.c_double(high)
    l, u = C.c_int64(), C.c_int64()
    out = np.empty(t.size, np.float32)
    _rc(load_library().thz_host_td_bandpass(t.ctypes.data, t.size, C.byref(lo), C.byref(hi), width,
                                            out.ctypes.data, C.byref(l), C.byref(u)), "td_bandpass")
    return out, lo.value, hi.value, l.value, u.value


def host_fd_bandpass(frequency, low: float, high: float, width: float):
    """-> (multiplier, lower_idx, upper_idx)"""
    f = np.ascontiguousarray(frequency, np.float32)
    l, u = C.c_int64(), C.c_int64()
    out = np.empty(f.size, np.float32)
    _rc(load_library().thz_host_fd_bandpass(f.ctypes.data, f.size, low, high, width, out.ctypes.data,
                                            C.byref(l), C.byref(u)), "fd_bandpass")
    return out, l.value, u.value


def host_water_line_mask(frequency, lines_thz, sigma_thz=0.01) -> np.ndarray:
    f = np.ascontiguousarray(frequency, np.float32)
    ln = np.ascontiguousarray(lines_thz, np.float32)
    out = np.empty(f.size, np.float32)
    _rc(load_library().thz_host_water_line_mask(f.ctypes.data, f.size, ln.ctypes.data, ln.size, sigma_thz,
                                                out.ctypes.data), "water_line_mask")
    return out


def host_wiener_filter(ref_fft, eps_rel=1e-3) -> np.ndarray:
    """ref_fft (nf, 2) f32 -> (nf, 2) complex multiplier"""
    r = np.ascontiguousarray(ref_fft, np.float32).reshape(-1, 2)
    out = np.empty_like(r)
    _rc(load_library().thz_host_wiener_filter(r.ctypes.data, r.shape[0], eps_rel, out.ctypes.data), "wiener_filter")
    return out


def host_tilt_plan(time, nx, ny, tilt_x_deg, tilt_y_deg, dx, dy):
    """-> (num_steps, new_time, insert_index (nx*ny int32))"""
    t = np.ascontiguousarray(time, np.float32)
    L = load_library()
    steps = int(L.thz_host_tilt_plan(t.ctypes.data, t.size, nx, ny, tilt_x_deg, tilt_y_deg, dx, dy, None, None))
    new_time = np.empty(t.size + 2 * steps, np.float32)
    ins = np.empty(nx * ny, np.int32)
    L.thz_host_tilt_plan(t.ctypes.data, t.size, nx, ny, tilt_x_deg, tilt_y_deg, dx, dy,
                         new_time.ctypes.data, ins.ctypes.data)
    return steps, new_time, ins


def host_psf_eval(psf: Psf, freqs):
    f = np.ascontiguousarray(freqs, np.float32)
    out = [np.empty(f.size, np.float32) for _ in range(4)]
    _rc(load_library().thz_host_psf_eval(C.byref(psf), f.ctypes.data, f.size, *[o.ctypes.data for o in out]), "psf_eval")
    return tuple(out)  # wx, wy, x0, y0


def host_filter_bank(time, cfg: DeconvCfg):
    t = np.ascontiguousarray(time, np.float32)
    filters = np.empty((cfg.n_filters, 499), np.float32)
    centers = np.empty(cfg.n_filters, np.float32)
    _rc(load_library().thz_host_filter_bank(t.ctypes.data, t.size, C.byref(cfg), filters.ctypes.data,
                                            centers.ctypes.data), "filter_bank")
    return filters, centers


def host_band_psf(psf: Psf, center_freq, dx, dy, img_rows, img_cols):
    r, c = C.c_size_t(), C.c_size_t()
    L = load_library()
    _rc(L.thz_host_band_psf(C.byref(psf), center_freq, dx, dy, img_rows, img_cols, None, C.byref(r), C.byref(c)), "band_psf")
    out = np.empty((r.value, c.value), np.float32)
    _rc(L.thz_host_band_psf(C.byref(psf), center_freq, dx, dy, img_rows, img_cols, out.ctypes.data,
                            C.byref(r), C.byref(c)), "band_psf")
    return out


def host_align_reference(scan_time, ref_time, ref_signal):
    """OpenRef alignment -> (aligned reference of the scan's length, mode 0/1/2)"""
    st, rt, rs = (np.ascontiguousarray(x, np.float32) for x in (scan_time, ref_time, ref_signal))
    out = np.empty(st.size, np.float32)
    mode = load_library().thz_host_align_reference(st.ctypes.data, st.size, rt.ctypes.data, rs.ctypes.data, rs.size,
                                                   out.ctypes.data)
    if mode < 0:
        raise ThzError(mode, "thz_host_align_reference")
    return out, mode


def host_optical_properties(sample_amp, sample_phase, ref_amp, ref_phase, freq, thickness):
    """-> (refractive index, absorption coefficient, extinction coefficient)"""
    arrs = [np.ascontiguousarray(x, np.float32) for x in (sample_amp, sample_phase, ref_amp, ref_phase, freq)]
    out = [np.empty(arrs[4].size, np.float32) for _ in range(3)]
    _rc(load_library().thz_host_optical_properties(*[a.ctypes.data for a in arrs], arrs[4].size, thickness,
                                                   *[o.ctypes.data for o in out]), "optical_properties")
    return tuple(out)


def voxel_cfg_default() -> VoxelCfg:
    cfg = VoxelCfg()
    _rc(load_library().thz_voxel_cfg_default(C.byref(cfg)), "voxel_cfg_default")
    return cfg


def host_gaussian_kernel1d(sigma, radius):
    out = np.empty(2 * radius + 1, np.float32)
    _rc(load_library().thz_host_gaussian_kernel1d(sigma, radius, out.ctypes.data), "gaussian_kernel1d")
    return out


def host_select_step(hist, k):
    """one level of the radix select -> (bin, rank inside the bin)"""
    h = np.ascontiguousarray(hist, np.uint64)
    b, r = C.c_int(), C.c_uint64()
    _rc(load_library().thz_host_select_step(h.ctypes.data, h.size, int(k), C.byref(b), C.byref(r)), "select_step")
    return b.value, r.value


def host_select_value(bin0, bin1, bin2) -> float:
    return float(load_library().thz_host_select_value(bin0, bin1, bin2))


def chain_cfg_default(time) -> ChainCfg:
    t = np.ascontiguousarray(time, np.float32)
    cfg = ChainCfg()
    _rc(load_library().thz_chain_cfg_default(t.ctypes.data, t.size, C.byref(cfg)), "chain_cfg_default")
    return cfg


def _pack_rois(polys):
    """list of (n, 2) vertex arrays -> (n_rois, size_t counts, u64 vertices) for thz_*_set_rois"""
    polys = [np.ascontiguousarray(p, np.uint64).reshape(-1, 2) for p in polys]
    counts = (_SZ * max(len(polys), 1))(*[p.shape[0] for p in polys])
    flat = np.concatenate(polys).ravel() if polys else np.zeros(0, np.uint64)
    return len(polys), counts, np.ascontiguousarray(flat, np.uint64)


def _roi_out(nt_out, want):
    nf = nt_out // 2 + 1
    sizes = dict(signal_fft=nf, phase_fft=nf, signal=nt_out, roi_data=nt_out)
    want = list(sizes) if want is None else want
    res = {k: np.empty(sizes[k], np.float32) for k in want}
    cnt = C.c_uint32()
    ro = RoiOut(count=C.addressof(cnt), **{k: v.ctypes.data for k, v in res.items()})
    return res, cnt, ro


class Session:
    """thz_session: resident cube + whole-chain recompute"""

    def __init__(self, eng: "Engine", nx, ny, time, dx=1.0, dy=1.0):
        self.eng, self.nx, self.ny = eng, nx, ny
        t = np.ascontiguousarray(time, np.float32)
        self.nt = t.size
        self.h = _P()
        eng._check(eng.lib.thz_session_create(eng.ctx, nx, ny, t.size, t.ctypes.data, dx, dy, C.byref(self.h)))

    def close(self):
        if self.h:
            self.eng.lib.thz_session_destroy(self.h)
            self.h = None

    def upload(self, cube, subtract_bias=True):
        """cube: host (nx, ny, nt) array, or None when THZ_BUF_RAW was filled on the device"""
        c = None if cube is None else np.ascontiguousarray(cube, np.float32)
        if c is not None and c.size != self.nx * self.ny * self.nt:
            raise ValueError(f"cube has {c.size} samples, the session {self.nx * self.ny * self.nt}")
        self.eng._check(self.eng.lib.thz_session_upload(self.h, c.ctypes.data if c is not None else None, int(subtract_bias)))

    def recompute(self, cfg: ChainCfg, start_stage: int = 1):
        """UpdateType::Filter(start_stage): 1 = everything, 6 / 7 = from the resident spectrum"""
        self.eng._check(self.eng.lib.thz_session_recompute_from(self.h, C.byref(cfg), int(start_stage)))

    def set_fd_filters(self, real_mask=None, cmask=None):
        """further Frequency-domain plugins: K14 (real, nf) and K13 (complex, (nf, 2))"""
        r = None if real_mask is None else np.ascontiguousarray(real_mask, np.float32)
        c = None if cmask is None else np.ascontiguousarray(cmask, np.float32)
        nf = r.size if r is not None else (c.size // 2 if c is not None else 0)
        self.eng._check(self.eng.lib.thz_session_set_fd_filters(self.h, r.ctypes.data if r is not None else None,
                                                                c.ctypes.data if c is not None else None, nf))

    def set_rois(self, polys):
        """regions of interest: list of (n, 2) integer vertex arrays (x, y) in raw-grid pixels; [] removes them"""
        n, counts, flat = _pack_rois(polys)
        self.eng._check(self.eng.lib.thz_session_set_rois(self.h, n, counts, flat.ctypes.data if flat.size else None))

    def roi(self, index, want=None):
        """per-region vectors of the last recompute -> dict (+ 'count')"""
        res, cnt, ro = _roi_out(self.nt_out, want)
        self.eng._check(self.eng.lib.thz_session_roi(self.h, index, C.byref(ro)))
        res["count"] = cnt.value
        return res

    def deconvolve(self, psf, cfg, abort=None, progress=None):
        """the chain's Deconvolution stage on the last recompute's output -> status (0 applied, 1 skipped)"""
        rc = self.eng.lib.thz_session_deconvolve(self.h, C.byref(psf), C.byref(cfg),
                                                 C.byref(abort) if abort is not None else None,
                                                 C.byref(progress) if progress is not None else None)
        if rc < 0:
            self.eng._check(rc)
        return rc

    @property
    def nt_out(self):
        return int(self.eng.lib.thz_session_nt_out(self.h))

    def grid(self):
        """(nx, ny, dx, dy) of the outputs: the raw grid, or the block grid behind a scaling stage"""
        nx, ny, dx, dy = _SZ(), _SZ(), C.c_float(), C.c_float()
        self.eng._check(self.eng.lib.thz_session_grid(self.h, C.byref(nx), C.byref(ny), C.byref(dx), C.byref(dy)))
        return nx.value, ny.value, dx.value, dy.value

    def time_out(self):
        t = np.empty(self.nt_out, np.float32)
        self.eng._check(self.eng.lib.thz_session_time_out(self.h, t.ctypes.data))
        return t

    def voxels(self, cfg: "VoxelCfg", max_instances=VOXEL_MAX_INSTANCES, scaling=1, orig_dims=None, capacity=None):
        """update_intensity_image's 3-D part -> (instances, threshold, (cube_width, cube_height, cube_depth))"""
        orig = orig_dims or (self.nx, self.ny, self.nt_out)
        n, thr = C.c_uint64(), C.c_float()
        dims = np.zeros(3, np.float32)
        if capacity is None:   # count first
            self.eng._check(self.eng.lib.thz_session_voxels(self.h, C.byref(cfg), max_instances, scaling, orig[0],
                                                            orig[1], orig[2], None, 0, C.byref(n), C.byref(thr),
                                                            dims.ctypes.data))
            capacity = n.value
        out = np.zeros(capacity, VOXEL_INSTANCE)
        self.eng._check(self.eng.lib.thz_session_voxels(self.h, C.byref(cfg), max_instances, scaling, orig[0], orig[1],
                                                        orig[2], out.ctypes.data if capacity else None, capacity,
                                                        C.byref(n), C.byref(thr), dims.ctypes.data))
        return out[:min(n.value, capacity)], thr.value, tuple(float(x) for x in dims)

    def plot(self, px, py, want=None):
        """UpdateType::Plot copy-out for pixel (px, py) -> dict of host vectors"""
        nto = self.nt_out
        nf = nto // 2 + 1
        sizes = dict(signal=self.nt, signal_fft=nf, phase_fft=nf, filtered_signal=nto, filtered_signal_fft=nf,
                     filtered_phase_fft=nf, avg_signal=nto, avg_signal_fft=nf, avg_phase_fft=nf)
        want = list(sizes) if want is None else want
        res = {k: np.empty(sizes[k], np.float32) for k in want}
        po = PlotOut(**{k: v.ctypes.data for k, v in res.items()})
        self.eng._check(self.eng.lib.thz_session_plot(self.h, px, py, C.byref(po)))
        return res

    def download(self, which, pix0=0, npix=None):
        nto = self.nt_out
        nf = nto // 2 + 1
        per = {BUF_RAW: (self.nt,), BUF_FFT: (nf, 2), BUF_AMPLITUDES: (nf,), BUF_PHASES: (nf,), BUF_DATA: (nto,),
               BUF_IMG: (), BUF_OPACITY: (nto,)}
        if which in per:
            gx, gy = (self.nx, self.ny) if which == BUF_RAW else self.grid()[:2]
            npix = gx * gy - pix0 if npix is None else npix
            out = np.empty((npix,) + per[which], np.float32)
        else:
            out = np.empty((nf, 2) if which == BUF_AVG_FFT else (nf,), np.float32)
            pix0, npix = 0, 1
        self.eng._check(self.eng.lib.thz_session_download(self.h, which, pix0, npix, out.ctypes.data))
        return out


GATHER_SMALL, GATHER_TIME, GATHER_ALL = range(3)
GROUP_ID_BYTES = 128


def host_slab(nx: int, world: int, rank: int):
    """(x0, n) rows of `rank`: the x-slab partition of include/thzgpu.h (thz_host_slab)"""
    x0, n = _SZ(), _SZ()
    _rc(load_library().thz_host_slab(nx, world, rank, C.byref(x0), C.byref(n)), "host_slab")
    return x0.value, n.value


def group_unique_id() -> bytes:
    buf = C.create_string_buffer(GROUP_ID_BYTES)
    _rc(load_library().thz_group_unique_id(buf), "group_unique_id")
    return buf.raw


class Group:
    """thz_group: the members of a multi-GPU tiling driven by this process.  Group(devices=[0, 1, ...]) is one
    process over n devices; Group(device=d, rank=r, world=w, uid=...) one member of a process-per-GPU launch."""

    def __init__(self, devices=None, device=None, rank=0, world=1, uid=None):
        self.lib = load_library()
        self.h = _P()
        if devices is not None:
            arr = (C.c_int * len(devices))(*devices)
            rc = self.lib.thz_group_create(arr, len(devices), C.byref(self.h))
        else:
            rc = self.lib.thz_group_create_rank(device, rank, world, uid, C.byref(self.h))
        if rc != THZ_OK:
            raise ThzError(rc, "thz_group_create failed (no HIP device, RCCL missing, or a bad device list)")
        self.world = self.lib.thz_group_world(self.h)
        self.ranks = [self.lib.thz_group_rank(self.h, i) for i in range(self.lib.thz_group_local_count(self.h))]

    def _check(self, rc):
        if rc != THZ_OK:
            raise ThzError(rc, self.lib.thz_group_last_error(self.h).decode())

    def engine(self, i: int) -> "Engine":
        """Engine view of local member i's context (owned by the group: do not close it)"""
        e = Engine.__new__(Engine)
        e.lib, e.ctx, e._bufs = self.lib, _P(self.lib.thz_group_ctx(self.h, i)), []
        return e

    def all_reduce_sum(self, bufs, count):
        arr = (_P * len(bufs))(*[_dp(b) for b in bufs])
        self._check(self.lib.thz_group_all_reduce_sum(self.h, arr, count))

    def all_reduce_u64(self, bufs, count):
        arr = (_P * len(bufs))(*[_dp(b) for b in bufs])
        self._check(self.lib.thz_group_all_reduce_u64(self.h, arr, count))

    def gather(self, send, counts, recv_root):
        arr = (_P * len(send))(*[_dp(b) for b in send])
        cnt = (_SZ * len(counts))(*counts)
        self._check(self.lib.thz_group_gather(self.h, arr, cnt, _dp(recv_root)))

    def sync(self):
        self._check(self.lib.thz_group_sync(self.h))

    def close(self):
        if self.h:
            self.lib.thz_group_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class GroupSession:
    """thz_group_session: one x-slab session per member, C2 / C1 inside the library"""

    def __init__(self, group: Group, nx, ny, time, dx=1.0, dy=1.0):
        self.g, self.nx, self.ny = group, nx, ny
        t = np.ascontiguousarray(time, np.float32)
        self.nt = t.size
        self.h = _P()
        group._check(group.lib.thz_group_session_create(group.h, nx, ny, t.size, t.ctypes.data, dx, dy, C.byref(self.h)))

    def member(self, i) -> "Session":
        """Session view of local member i's slab session (owned by the group session: do not close it)"""
        s = Session.__new__(Session)
        x0, n = host_slab(self.nx, self.g.world, self.g.ranks[i])
        s.eng, s.nx, s.ny, s.nt = self.g.engine(i), n, self.ny, self.nt
        s.h = _P(self.g.lib.thz_group_session_member(self.h, i))
        return s

    def member_buffer(self, i, which) -> int:
        """device pointer of a buffer of local member i's slab session"""
        return self.g.lib.thz_session_buffer(_P(self.g.lib.thz_group_session_member(self.h, i)), which) or 0

    def upload(self, cube=None, subtract_bias=True):
        c = None if cube is None else np.ascontiguousarray(cube, np.float32)
        self.g._check(self.g.lib.thz_group_session_upload(self.h, c.ctypes.data if c is not None else None, int(subtract_bias)))

    def recompute(self, cfg: ChainCfg, start_stage=1, gather=GATHER_SMALL):
        self.g._check(self.g.lib.thz_group_session_recompute(self.h, C.byref(cfg), int(start_stage), int(gather)))

    def set_rois(self, polys):
        """regions of interest over the whole grid (every rank sets the same ones)"""
        n, counts, flat = _pack_rois(polys)
        self.g._check(self.g.lib.thz_group_session_set_rois(self.h, n, counts, flat.ctypes.data if flat.size else None))

    def roi(self, index, want=None, nt_out=None):
        res, cnt, ro = _roi_out(self.nt if nt_out is None else nt_out, want)
        self.g._check(self.g.lib.thz_group_session_roi(self.h, index, C.byref(ro)))
        res["count"] = cnt.value
        return res

    def deconvolve(self, psf, cfg, abort=None, progress=None):
        """band-parallel Deconvolution stage over the group -> status (0 applied, 1 skipped)"""
        rc = self.g.lib.thz_group_session_deconvolve(self.h, C.byref(psf), C.byref(cfg),
                                                     C.byref(abort) if abort is not None else None,
                                                     C.byref(progress) if progress is not None else None)
        if rc < 0:
            self.g._check(rc)
        return rc

    def grid(self):
        """(nx, ny) of the outputs' whole grid: the raw one, or the block grid behind a scaling stage"""
        nx, ny = _SZ(), _SZ()
        self.g._check(self.g.lib.thz_group_session_grid(self.h, C.byref(nx), C.byref(ny)))
        return nx.value, ny.value

    def download(self, which, nt_out=None):
        """gathered buffer of rank 0 (whole grid) or a pixel-mean vector"""
        nto = self.nt if nt_out is None else nt_out
        nf = nto // 2 + 1
        per = {BUF_FFT: (nf, 2), BUF_AMPLITUDES: (nf,), BUF_PHASES: (nf,), BUF_DATA: (nto,), BUF_IMG: ()}
        if which in per:
            gx, gy = self.grid()
            out = np.empty((gx * gy,) + per[which], np.float32)
            self.g._check(self.g.lib.thz_group_session_download(self.h, which, 0, gx * gy, out.ctypes.data))
        else:
            out = np.empty((nf, 2) if which == BUF_AVG_FFT else (nf,), np.float32)
            self.g._check(self.g.lib.thz_group_session_download(self.h, which, 0, 1, out.ctypes.data))
        return out

    def close(self):
        if self.h:
            self.g.lib.thz_group_session_destroy(self.h)
            self.h = None


class DevBuf:
    """A device allocation owned through thz_malloc/thz_free."""

    def __init__(self, eng: "Engine", nbytes: int):
        self.eng = eng
        self.nbytes = int(nbytes)
        p = _P()
        eng._check(eng.lib.thz_malloc(eng.ctx, C.byref(p), self.nbytes))
        self.ptr = p.value

    def free(self):
        if self.ptr:
            self.eng.lib.thz_free(self.eng.ctx, self.ptr)
            self.ptr = None

    def upload(self, a: np.ndarray) -> "DevBuf":
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        self.eng._check(self.eng.lib.thz_memcpy_h2d(self.eng.ctx, self.ptr, a.ctypes.data, a.nbytes))
        return self

    def download(self, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        self.eng._check(self.eng.lib.thz_memcpy_d2h(self.eng.ctx, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def zero(self):
        self.eng._check(self.eng.lib.thz_memset(self.eng.ctx, self.ptr, 0, self.nbytes))
        return self


def _dp(x) -> Optional[int]:
    """device pointer of a DevBuf / int / None"""
    if x is None:
        return None
    if isinstance(x, DevBuf):
        return x.ptr
    return int(x)


class Engine:
    """One thz_ctx.  Methods are 1:1 with the C entry points."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        self.ctx = _P()
        rc = self.lib.thz_create(device, C.byref(self.ctx))
        if rc != THZ_OK:
            raise ThzError(rc, "thz_create failed (no HIP device?)")
        self._bufs = []

    # -- helpers
    def _check(self, rc: int):
        if rc != THZ_OK:
            raise ThzError(rc, self.lib.thz_last_error(self.ctx).decode())

    def release_scratch(self):
        """frees the device scratch the context keeps between calls (thz_release_scratch)"""
        self._check(self.lib.thz_release_scratch(self.ctx))

    def close(self):
        if self.ctx:
            for b in self._bufs:
                b.free()
            self._bufs = []
            self.lib.thz_destroy(self.ctx)
            self.ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def alloc(self, nbytes: int) -> DevBuf:
        b = DevBuf(self, nbytes)
        self._bufs.append(b)
        return b

    def to_device(self, a: np.ndarray) -> DevBuf:
        a = np.ascontiguousarray(a)
        return self.alloc(max(a.nbytes, 4)).upload(a)

    def empty(self, shape, dtype=np.float32) -> DevBuf:
        return self.alloc(max(int(np.prod(shape)) * np.dtype(dtype).itemsize, 4))

    def sync(self):
        self._check(self.lib.thz_sync(self.ctx))

    @property
    def stream(self) -> int:
        return self.lib.thz_stream(self.ctx) or 0

    # -- plan
    def set_time_axis(self, time: np.ndarray):
        t = np.ascontiguousarray(time, np.float32)
        self._check(self.lib.thz_set_time_axis(self.ctx, t.ctypes.data, t.size))
        self.nt = int(self.lib.thz_nt(self.ctx))
        self.nf = int(self.lib.thz_nf(self.ctx))

    def set_kernel_family(self, family: int):
        """0 = auto (F kernels where available), 1 = G kernels everywhere"""
        self._check(self.lib.thz_set_kernel_family(self.ctx, int(family)))

    def frequency(self) -> np.ndarray:
        f = np.empty(self.nf, np.float32)
        self._check(self.lib.thz_get_frequency(self.ctx, f.ctypes.data))
        return f

    def kernel_variant(self) -> str:
        return self.lib.thz_kernel_variant(self.ctx).decode()

    # -- stages
    def fft(self, npix, d_in, win_a=None, win_b=None, data_out=None, fft=None, amp=None,
            phase=None, fd_mask=None):
        self._check(self.lib.thz_fft(self.ctx, npix, _dp(d_in), _dp(win_a), _dp(win_b),
                                     _dp(data_out), _dp(fft), _dp(amp), _dp(phase), _dp(fd_mask)))

    def apply_fd_mask(self, npix, fft, amp, mask):
        self._check(self.lib.thz_apply_fd_mask(self.ctx, npix, _dp(fft), _dp(amp), _dp(mask)))

    def apply_fd_cmask(self, npix, fft, amp, cmask):
        self._check(self.lib.thz_apply_fd_cmask(self.ctx, npix, _dp(fft), _dp(amp), _dp(cmask)))

    def ifft(self, npix, fft, td_win, data_out, img=None):
        self._check(self.lib.thz_ifft(self.ctx, npix, _dp(fft), _dp(td_win), _dp(data_out), _dp(img)))

    def pipeline_ex(self, npix, raw, pre_win, fd_mask, fd_cmask, post_win, fft, amp, phase, data_out, img=None, sums=None, band=None):
        """thz_pipeline_ex: the fused chain with a complex per-bin multiplier and / or in-launch pixel sums; band = (lo, hi):
        the bins outside are zero in fd_mask (thz_host_fd_bandpass's index range)"""
        io = PipelineIo(*[_dp(x) for x in (raw, pre_win, fd_mask, fd_cmask, post_win, fft, amp, phase, data_out, img, sums)],
                        *(band if band else (0, 0)))
        self._check(self.lib.thz_pipeline_ex(self.ctx, npix, C.byref(io)))

    def pipeline(self, npix, raw, pre_win, fd_mask, post_win, fft, amp, phase, data_out, img):
        self._check(self.lib.thz_pipeline(self.ctx, npix, _dp(raw), _dp(pre_win), _dp(fd_mask),
                                          _dp(post_win), _dp(fft), _dp(amp), _dp(phase),
                                          _dp(data_out), _dp(img)))

    def apply_td_window(self, npix, d_in, win, d_out):
        self._check(self.lib.thz_apply_td_window(self.ctx, npix, _dp(d_in), _dp(win), _dp(d_out)))

    def intensity(self, npix, data, img):
        self._check(self.lib.thz_intensity(self.ctx, npix, _dp(data), _dp(img)))

    def subtract_bias(self, npix, data, img=None):
        self._check(self.lib.thz_subtract_bias(self.ctx, npix, _dp(data), _dp(img)))

    def pixel_mean(self, nx, ny, length, ncomp, arr, out):
        self._check(self.lib.thz_pixel_mean(self.ctx, nx, ny, length, ncomp, _dp(arr), _dp(out)))

    def pixel_sum(self, npix, length, ncomp, arr, out):
        self._check(self.lib.thz_pixel_sum(self.ctx, npix, length, ncomp, _dp(arr), _dp(out)))

    def roi_mask(self, poly_xy, scaling, shape0, shape1, mask):
        poly = np.ascontiguousarray(poly_xy, np.uint64).reshape(-1, 2)
        self._check(self.lib.thz_roi_mask(self.ctx, poly.ctypes.data if poly.size else None,
                                          poly.shape[0], scaling, shape0, shape1, _dp(mask)))

    def roi_mean(self, arr, shape0, shape1, length, mask, out, count=None, sum_only=False):
        self._check(self.lib.thz_roi_mean(self.ctx, _dp(arr), shape0, shape1, length, _dp(mask),
                                          _dp(out), _dp(count), int(sum_only)))

    def scale3d(self, arr, nx, ny, length, ncomp, s, out):
        self._check(self.lib.thz_scale3d(self.ctx, _dp(arr), nx, ny, length, ncomp, s, _dp(out)))

    def tilt_apply(self, npix, d_in, nt_in, d_taper, d_insert, nt_out, d_out):
        self._check(self.lib.thz_tilt_apply(self.ctx, npix, _dp(d_in), nt_in, _dp(d_taper), _dp(d_insert),
                                            nt_out, _dp(d_out)))

    def polar_ifft(self, amp, phase, zero_dc_imag=False):
        """ifft's avg_in_fourier_space branch: from_polar(avg amplitude, avg phase) -> C2R / nt (host vectors)"""
        a, p = (np.ascontiguousarray(x, np.float32) for x in (amp, phase))
        out = np.empty(self.nt, np.float32)
        self._check(self.lib.thz_polar_ifft(self.ctx, a.ctypes.data, p.ctypes.data, int(zero_dc_imag), out.ctypes.data))
        return out

    def deconvolve(self, psf: Psf, cfg: DeconvCfg, nx, ny, dx, dy, d_in, d_out, d_img=None, d_gains=None,
                   abort=None, progress=None):
        """-> status (0 applied, 1 skipped by one of the reference's guards).  abort: ctypes.c_int polled
        between iteration batches (the reference's AtomicBool); progress: ctypes.c_float written 0..1"""
        rc = self.lib.thz_deconvolve(self.ctx, C.byref(psf), C.byref(cfg), nx, ny, dx, dy, _dp(d_in), _dp(d_out),
                                     _dp(d_img), _dp(d_gains), C.byref(abort) if abort is not None else None,
                                     C.byref(progress) if progress is not None else None)
        if rc < 0:
            self._check(rc)
        return rc

    def reference_spectrum(self, scan_time, ref_time, ref_signal, window_type=0, lower=1.0, upper=7.0):
        """OpenRef -> (aligned + windowed reference, amplitudes, unwrapped phases)"""
        st, rt, rs = (np.ascontiguousarray(x, np.float32) for x in (scan_time, ref_time, ref_signal))
        nf = st.size // 2 + 1
        ref, amp, ph = np.empty(st.size, np.float32), np.empty(nf, np.float32), np.empty(nf, np.float32)
        w = WindowCfg(window_type, lower, upper)
        self._check(self.lib.thz_reference_spectrum(self.ctx, st.ctypes.data, st.size, rt.ctypes.data, rs.ctypes.data,
                                                    rs.size, C.byref(w), ref.ctypes.data, amp.ctypes.data,
                                                    ph.ctypes.data))
        return ref, amp, ph

    def voxel_opacity(self, npix, nt, d_data, cfg: VoxelCfg, d_opacity):
        self._check(self.lib.thz_voxel_opacity(self.ctx, npix, nt, _dp(d_data), C.byref(cfg), _dp(d_opacity)))

    def select_histogram(self, d_vals, n, level, prefix, d_hist):
        self._check(self.lib.thz_select_histogram(self.ctx, _dp(d_vals), n, level, prefix, _dp(d_hist)))

    def kth_largest(self, d_vals, n, k) -> float:
        v = C.c_float()
        self._check(self.lib.thz_kth_largest(self.ctx, _dp(d_vals), n, k, C.byref(v)))
        return v.value

    def voxel_threshold(self, d_opacity, n, max_instances=VOXEL_MAX_INSTANCES) -> float:
        v = C.c_float()
        self._check(self.lib.thz_voxel_threshold(self.ctx, _dp(d_opacity), n, max_instances, C.byref(v)))
        return v.value

    def voxel_instances(self, d_opacity, gw, gh, gd, threshold, time_span, scaling, orig_dims, d_out, capacity,
                        x0=0, gw_total=None):
        """-> (count, (cube_width, cube_height, cube_depth))"""
        n = C.c_uint64()
        dims = np.zeros(3, np.float32)
        self._check(self.lib.thz_voxel_instances(self.ctx, _dp(d_opacity), gw, gh, gd, x0,
                                                 gw if gw_total is None else gw_total, threshold, time_span,
                                                 scaling, orig_dims[0], orig_dims[1], orig_dims[2], _dp(d_out),
                                                 capacity, C.byref(n), dims.ctypes.data))
        return n.value, tuple(float(x) for x in dims)

    def traffic_probe(self, npix, nt, d_in, d_fft, d_amp, d_phase, d_out):
        self._check(self.lib.thz_traffic_probe(self.ctx, npix, nt, _dp(d_in), _dp(d_fft), _dp(d_amp), _dp(d_phase),
                                               _dp(d_out)))

    def synth_cube(self, d_out, ntraces, first_trace, d_time, seed=0x7A3D2026, subtract_bias=True):
        self._check(self.lib.thz_synth_cube(self.ctx, _dp(d_out), ntraces, first_trace, _dp(d_time),
                                            seed, int(subtract_bias)))

    def enable_timing(self, mode=1):
        self._check(self.lib.thz_enable_timing(self.ctx, int(mode)))

    def timing_collect(self, stage: int):
        """-> (total_ns, calls) of `stage` since the last collect (deferred mode)"""
        t, n = C.c_uint64(), C.c_uint64()
        self._check(self.lib.thz_timing_collect(self.ctx, stage, C.byref(t), C.byref(n)))
        return t.value, n.value

    def stage_time_ns(self, stage: int) -> int:
        v = C.c_uint64()
        self._check(self.lib.thz_stage_time_ns(self.ctx, stage, C.byref(v)))
        return v.value
