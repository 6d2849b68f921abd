/* thzgpu.h — C ABI of libthzgpu.so, the MI355X (gfx950) engine for the
 * data_thread recompute path of unibe-icelab/thz-image-explorer.
 *
 * This is the drop-in boundary: plain pointers and sizes, no
 * C++/torch types.  Each entry point names the reference interface it
 * replaces (paths relative to the reference checkout).  A Rust maintainer
 * binds these with an `extern "C"` block (INTEGRATION.md shows the stub); the
 * tests bind them with ctypes; the C++ host mirror (thz_image_explorer_amd/host)
 * sits on top of them.
 *
 * Conventions
 *  - One context per process per GPU; calls on one context are made from one
 *    thread at a time (the reference's single data thread, data_thread.rs:162).
 *  - Every function returns THZ_OK (0) or a negative thz_status; it never
 *    throws or aborts.  thz_last_error() gives the text.  On error device
 *    buffers named as outputs are unspecified but inputs are untouched, so a
 *    caller can do what the reference does on failure: keep `input.clone()`.
 *  - Pointers named d_* are DEVICE pointers (from thz_malloc or any other HIP
 *    allocation on the context's device); all others are host pointers.
 *  - Layout is the reference's: cubes are C-order (x, y, t|f) with the last
 *    axis contiguous (data_container.rs:136-151); complex = interleaved
 *    {re, im} f32 (num_complex::Complex32).  `npix` = nx*ny traces.
 *  - Kernels are enqueued on the context's stream; thz_sync() / any D2H copy
 *    waits for them.
 */
#ifndef THZGPU_H
#define THZGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define THZGPU_ABI_VERSION 2

typedef struct thz_ctx thz_ctx;

typedef enum thz_status {
    THZ_OK = 0,
    THZ_ERR_INVALID = -1,      /* bad argument / shape */
    THZ_ERR_UNSUPPORTED = -2,  /* transform length not supported */
    THZ_ERR_HIP = -3,          /* HIP runtime failure (no GPU, OOM, fault) */
    THZ_ERR_NOT_READY = -4,    /* geometry / cube not set */
    THZ_ERR_ABORTED = -5,      /* abort flag observed between kernel batches */
    THZ_SKIPPED = 1            /* not an error: a guard of the reference applied, output = input */
} thz_status;

/* FftWindowType, math_tools.rs:35-46 (same order) */
typedef enum thz_window_type {
    THZ_WIN_ADAPTED_BLACKMAN = 0,
    THZ_WIN_BLACKMAN = 1,
    THZ_WIN_HANNING = 2,
    THZ_WIN_HAMMING = 3,
    THZ_WIN_FLAT_TOP = 4
} thz_window_type;

/* ConfigContainer.fft_window / fft_window_type, config.rs:171-213 */
typedef struct thz_window_cfg {
    int32_t type;   /* thz_window_type */
    float lower;    /* fft_window[0], ps; default 1.0 */
    float upper;    /* fft_window[1], ps; default 7.0 */
} thz_window_cfg;

/* ------------------------------------------------------------------ */
/* lifecycle                                                           */
/* ------------------------------------------------------------------ */

/* Creates a context on HIP device `device`.  Fails with THZ_ERR_HIP when no
 * GPU is present — there is no CPU fallback. */
int thz_create(int device, thz_ctx **out);
void thz_destroy(thz_ctx *ctx);
/* Frees the device scratch the context keeps between calls (the pixel-sum workspace, the blocks of the last
 * thz_deconvolve — see there); the next call that needs them allocates again.  Waits for the context's stream.
 * Not in the reference (its buffers die with each filter call): a host under memory pressure calls this after a
 * Deconvolution, the way it would drop a cache. */
int thz_release_scratch(thz_ctx *ctx);
const char *thz_last_error(const thz_ctx *ctx);
int thz_abi_version(void);
/* hipStream_t of the context as an opaque pointer (for callers that want to
 * order their own HIP work, e.g. torch.cuda.ExternalStream). */
void *thz_stream(thz_ctx *ctx);
int thz_sync(thz_ctx *ctx);

/* device memory helpers so that a non-HIP caller can own buffers */
int thz_malloc(thz_ctx *ctx, void **d_ptr, size_t bytes);
int thz_free(thz_ctx *ctx, void *d_ptr);
int thz_memcpy_h2d(thz_ctx *ctx, void *d_dst, const void *src, size_t bytes);
int thz_memcpy_d2h(thz_ctx *ctx, void *dst, const void *d_src, size_t bytes);
int thz_memcpy_d2d(thz_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);
int thz_memset(thz_ctx *ctx, void *d_dst, int value, size_t bytes);

/* ------------------------------------------------------------------ */
/* geometry / plan                                                     */
/* ------------------------------------------------------------------ */

/* Sets trace length and time axis; builds the transform tables ("plans").
 * Replaces RealFftPlanner::plan_fft_forward/inverse + the frequency-axis rule
 * at io.rs:614-621, data_thread.rs:1194-1207, tilt_compensation.rs:206-217:
 * frequency[i] = i / (time[nt-1] - time[0]), i = 0..nt/2.
 * Supported nt: any length 2..65536, as realfft plans any length.  Lengths whose transform buffers do not fit a
 * CU's LDS — not a power of two above 8191 (4096 with thz_set_kernel_family(1)), powers of two above 16384 — run
 * with those buffers in global scratch (up to 1 GiB, allocated here): correct, two orders of magnitude slower per
 * sample than the lengths with a kernel of their own. */
int thz_set_time_axis(thz_ctx *ctx, const float *time, size_t nt);
/* Kernel family selection: 0 = automatic (register-resident three-pass "F"
 * kernels for nt = 1024/2048/4096; mixed-radix "P" kernels for the lengths that
 * factor into three small radices — nt = 1001 = 7 x 11 x 13, the length of the
 * reference's real scans, and 1000 / 1200 / 1500 / 2000; half-length mixed-radix "PH" kernels for
 * 2002 / 2400 / 3000 / 4000 (twice a P length); chirp-z over the F core — "FB" kernels —
 * for the other lengths that are not a power of two; LDS Stockham "G" kernels
 * for the remaining powers of two), 1 = G kernels (Stockham / Bluestein in LDS)
 * for every length, 2 = automatic without the P kernels (A/B measurements, tests).
 * Re-plans if a time axis is already set. */
int thz_set_kernel_family(thz_ctx *ctx, int family);
size_t thz_nt(const thz_ctx *ctx);
size_t thz_nf(const thz_ctx *ctx);
int thz_get_frequency(const thz_ctx *ctx, float *frequency /* nf */);

/* ------------------------------------------------------------------ */
/* host-side multiplier vectors (O(nt) work, no GPU involved)          */
/* ------------------------------------------------------------------ */
/* The reference re-evaluates its window formulas for every trace inside the
 * pixel loops; they only depend on the axis, so the engine evaluates them once
 * per call on the host, in the reference's f32 operation order, and the
 * kernels multiply by the resulting vector.  These need no context. */

/* frequency[i] = i / (time[nt-1] - time[0]), i = 0..nt/2 (io.rs:614-621). */
int thz_host_frequency_axis(const float *time, size_t nt, float *frequency /* nt/2+1 */);

/* Window multiplier of the fft stage, math_tools.rs:102-198, 356-371:
 * out[i] = w(time[i]) such that windowed = data * out. */
int thz_host_fft_window(const float *time, size_t nt, const thz_window_cfg *cfg,
                        float *out /* nt */);

/* apply_adapted_blackman_window on ones (math_tools.rs:102-122) over an
 * arbitrary axis slice; tilt compensation's tail taper is (lower 0, upper 7),
 * tilt_compensation.rs:186-188. */
int thz_host_adapted_blackman(const float *axis, size_t len, float lower, float upper,
                              float *out /* len */);

/* "Time Band Pass" multiplier, band_pass_td_before_fft.rs:124-182 (and
 * _after_fft.rs): 0 outside [lower,upper), adapted-Blackman taper inside.
 * The low / high values are clamped in place exactly as the filter clamps its
 * own fields (:137-138).  lower/upper indices are returned when non-NULL. */
int thz_host_td_bandpass(const float *time, size_t nt, double *low, double *high,
                         double window_width, float *out /* nt */, int64_t *lower,
                         int64_t *upper);

/* "Frequency Band Pass" multiplier, band_pass_fd.rs:135-168 + zero padding
 * :194-212: 0 outside [lower,upper), taper inside. */
int thz_host_fd_bandpass(const float *frequency, size_t nf, double low, double high,
                         double window_width, float *out /* nf */, int64_t *lower,
                         int64_t *upper);

/* Build-defined frequency-domain filters (the reference has neither; it only
 * draws the water lines, gui/center_panel.rs:477-485).  Both
 * produce per-bin multipliers for thz_apply_fd_mask / thz_apply_fd_cmask.
 *   water lines (K14): out[k] = prod_i (1 - exp(-((f_k - line_i)/sigma)^2))
 *   Wiener (K13): out[k] = conj(R[k]) / (|R[k]|^2 + eps_rel * max_j |R[j]|^2),
 *                 R = spectrum of the reference pulse (interleaved, nf bins) */
int thz_host_water_line_mask(const float *frequency, size_t nf, const float *lines_thz,
                             size_t n_lines, float sigma_thz, float *out /* nf */);
int thz_host_wiener_filter(const float *ref_fft, size_t nf, float eps_rel,
                           float *out_cmask /* nf interleaved complex */);

/* TiltCompensation::filter geometry, tilt_compensation.rs:104-175 (K11): the
 * time extension (floor((|cx*tx| + |cy*ty|)/c/0.05)*0.05 with the reference's
 * hard-coded dt = 0.05 ps), the extended time axis (front/back linspaces) and,
 * per pixel, the index at which the tapered trace is inserted.  Returns
 * num_steps; the new length is nt + 2*num_steps.  new_time / insert_index may
 * be NULL (sizing call). */
size_t thz_host_tilt_plan(const float *time, size_t nt, size_t nx, size_t ny, double tilt_x_deg,
                          double tilt_y_deg, float dx, float dy,
                          float *new_time /* nt + 2*num_steps */,
                          int32_t *insert_index /* nx*ny */);

/* ------------------------------------------------------------------ */
/* stage kernels (device pointers)                                     */
/* ------------------------------------------------------------------ */

/* math_tools::fft, math_tools.rs:330-398 (K1+K2+K3), optionally fused with
 * the Frequency Band Pass multiply, band_pass_fd.rs:184-187 (K4).
 *   d_in        (npix, nt) f32
 *   d_win_a/b   (nt) f32 multipliers applied in that order, each may be NULL
 *   d_data_out  (npix, nt) windowed trace (the stage's `data` output), or NULL
 *   d_fft       (npix, nf) complex, or NULL
 *   d_amp       (npix, nf) = |X| (Complex::norm, :384), or NULL
 *   d_phase     (npix, nf) = numpy_unwrap(arg X, 2*pi) (:387-388), or NULL
 *   d_fd_mask   (nf) f32 or NULL: when given, d_fft and d_amp are multiplied
 *               by it (phases are not — band_pass_fd.rs leaves them) */
int thz_fft(thz_ctx *ctx, size_t npix, const float *d_in, const float *d_win_a,
            const float *d_win_b, float *d_data_out, float *d_fft, float *d_amp, float *d_phase,
            const float *d_fd_mask);

/* FrequencyDomainBandPass::filter per-pixel part, band_pass_fd.rs:175-212,
 * in place: fft *= mask, amp *= mask.  Either array may be NULL. */
int thz_apply_fd_mask(thz_ctx *ctx, size_t npix, float *d_fft, float *d_amp, const float *d_mask);

/* Same with a complex per-bin multiplier (interleaved, nf entries): the
 * build-defined reference-pulse (Wiener) deconvolution K13.  Bin 0 and, for
 * even nt, the last bin have their imaginary part forced to 0 so the result
 * stays a valid C2R input (SURVEY a'-4).  amp *= |mask|. */
int thz_apply_fd_cmask(thz_ctx *ctx, size_t npix, float *d_fft, float *d_amp,
                       const float *d_cmask);

/* math_tools::ifft per-pixel part, math_tools.rs:545-568 (K5: C2R then /nt),
 * optionally fused with a time multiplier (K6, band_pass_td_after_fft.rs) and
 * the intensity image (K7, data_thread.rs:1288-1307).
 *   d_fft      (npix, nf) complex
 *   d_td_win   (nt) or NULL
 *   d_data_out (npix, nt)
 *   d_img      (npix) or NULL: sum_t out^2 */
int thz_ifft(thz_ctx *ctx, size_t npix, const float *d_fft, const float *d_td_win,
             float *d_data_out, float *d_img);

/* The avg_in_fourier_space branch of math_tools::ifft (math_tools.rs:442-470) and
 * its per-ROI twin (:496-529): spectrum[k] = from_polar(amp[k], phase[k]) of an
 * AVERAGED amplitude / unwrapped-phase pair, C2R, / nt — one trace, so host
 * vectors in and out (amp, phase: nf; out: nt).  zero_dc_imag: the ROI branch
 * clears Im(spectrum[0]) by hand (:510-512).  The transform ignores the
 * imaginary parts of bin 0 and (even nt) the last bin, as realfft's does. */
int thz_polar_ifft(thz_ctx *ctx, const float *amp, const float *phase, int zero_dc_imag, float *out);

/* Whole default chain for one cube tile in ONE launch (the hot path):
 * raw -> *pre window -> R2C -> amp/phase/unwrap -> *fd mask -> store
 *     -> C2R /nt -> *post window -> store + intensity.
 * d_pre_win is the composition of every time-domain multiplier in front of
 * the transform (tilt taper, Time Band Pass, fft window); d_post_win the Time
 * Band Pass after the inverse.  Output pointers other than d_data_out may be
 * NULL (not materialised). */
int thz_pipeline(thz_ctx *ctx, size_t npix, const float *d_raw, const float *d_pre_win,
                 const float *d_fd_mask, const float *d_post_win, float *d_fft, float *d_amp,
                 float *d_phase, float *d_data_out, float *d_img);

/* The same launch with every optional input and output of the fused chain:
 *   d_fd_cmask  (nf) interleaved complex per-bin multiplier or NULL — the build-defined
 *               reference-pulse (Wiener) filter K13 (thz_host_wiener_filter) inside the
 *               ONE pass over HBM: spectrum = X * (cmask * mask), imaginary parts of bin 0 and (even
 *               nt) of the last bin forced to 0 (math_tools.rs:510-512), amplitudes = |X cmask mask|,
 *               phases those of X (band_pass_fd.rs:184-212 does not touch them either).  Fused for nt = 1024 / 2048 /
 *               4096 and 1001 / 1000 / 1200 / 1500 / 2000 (the F and P kernel families); trace lengths
 *               without a fused kernel for it run fft -> thz_apply_fd_cmask -> ifft internally.
 *   d_sums      (2 nf) or NULL: sum over the npix traces of the stored amplitudes [0, nf) and of the
 *               unwrapped phases [nf, 2 nf) — the numerators of the pixel means of the ifft stage
 *               (math_tools.rs:427-440; divide by nx ny, or all-reduce the sums of the tiles first).  For
 *               nt = 1024 / 2048 / 4096 and 1001 / 1000 / 1200 / 1500 / 2000 they are taken INSIDE the launch (every block adds its
 *               traces' values to accumulators in LDS, wave by wave in a fixed order; a small pass adds the
 *               blocks' rows), for other lengths by thz_pixel_sum passes over d_amp / d_phase behind it.
 *               Deterministic; the summation order differs from the reference's sequential one (<= 2e-6
 *               relative); thz_pixel_mean is the bit-exact form.
 * d_fft, d_amp, d_phase and d_data_out are required here. */
typedef struct thz_pipeline_io {
    const float *d_raw;      /* (npix, nt) */
    const float *d_pre_win;  /* (nt) or NULL */
    const float *d_fd_mask;  /* (nf) real or NULL */
    const float *d_fd_cmask; /* (nf) complex or NULL */
    const float *d_post_win; /* (nt) or NULL */
    float *d_fft;            /* (npix, nf) complex */
    float *d_amp;            /* (npix, nf) */
    float *d_phase;          /* (npix, nf) */
    float *d_data_out;       /* (npix, nt) */
    float *d_img;            /* (npix) or NULL */
    float *d_sums;           /* (2 nf) or NULL */
    /* Optional: the caller's word that d_fd_mask is ZERO at every bin outside [band_lo, band_hi) — the index range
     * thz_host_fd_bandpass hands back (lower, upper).  0, 0: unknown.  With d_fd_cmask and d_sums at nt = 4096 the fused
     * kernel then stages only the band's bins of the complex multiplier (half the table at the default 0.2-5 THz) and
     * keeps its eighth wave per block; results are the same bit for bit.  A wrong range gives wrong spectra. */
    size_t band_lo, band_hi;
} thz_pipeline_io;
int thz_pipeline_ex(thz_ctx *ctx, size_t npix, const thz_pipeline_io *io);

/* Time multiplier alone (K6): out = in * win; in place allowed. */
int thz_apply_td_window(thz_ctx *ctx, size_t npix, const float *d_in, const float *d_win,
                        float *d_out);

/* Intensity image (K7), data_thread.rs:1288-1307 / io.rs:588-594. */
int thz_intensity(thz_ctx *ctx, size_t npix, const float *d_data, float *d_img);

/* Load-time preprocessing, io.rs:578-596: per-trace subtract data[x,y,0],
 * then intensity.  In place; d_img may be NULL. */
int thz_subtract_bias(thz_ctx *ctx, size_t npix, float *d_data, float *d_img);

/* Pixel means (K8), math_tools.rs:421-440: mean over x then over y of an
 * (nx, ny, len*ncomp) f32 array -> len*ncomp values.  ncomp = 2 for complex.
 * Also used for avg_data (data_thread.rs:1423-1429).
 * d_out holds *sums* scaled as the reference does: (sum_x / nx) summed over y
 * / ny.  For multi-GPU tiles use thz_pixel_sum and divide after the
 * all-reduce. */
int thz_pixel_mean(thz_ctx *ctx, size_t nx, size_t ny, size_t len, int ncomp, const float *d_arr,
                   float *d_out);
int thz_pixel_sum(thz_ctx *ctx, size_t npix, size_t len, int ncomp, const float *d_arr,
                  float *d_out);

/* ROI (K9), math_tools.rs:574-661.  Polygon vertices are (x, y) pairs of
 * u64 as the reference's Vec<(usize, usize)>; arithmetic is u64 with
 * release-mode wrapping, bit-exact.  mask is (shape0, shape1) u8 indexed
 * [y * shape1 + x] in the function's own (swapped) coordinates. */
int thz_roi_mask(thz_ctx *ctx, const uint64_t *poly_xy, size_t n_vertices, uint64_t scaling,
                 size_t shape0, size_t shape1, uint8_t *d_mask);
/* mean over masked pixels of data[shape0 - y - 1, x, :] (:647); zeros when
 * the mask is empty (:656-658).  d_count (u32, may be NULL) gets the pixel
 * count; with sum_only != 0 the division is skipped (multi-GPU partials). */
int thz_roi_mean(thz_ctx *ctx, const float *d_arr, size_t shape0, size_t shape1, size_t len,
                 const uint8_t *d_mask, float *d_out, uint32_t *d_count, int sum_only);

/* math_tools::scaling scale_3d helper, math_tools.rs:273-301 (K10). */
int thz_scale3d(thz_ctx *ctx, const float *d_arr, size_t nx, size_t ny, size_t len, int ncomp,
                size_t s, float *d_out);

/* TiltCompensation::filter per-pixel copy, tilt_compensation.rs:171-201 (K11):
 * out[p, :ins] = in[p, 0]; out[p, ins:ins+nt_in] = in[p, :] * taper (clipped at
 * nt_out); zeros behind.  d_taper = thz_host_adapted_blackman(time, 0, 7).
 * The caller then re-plans with thz_set_time_axis(new_time) exactly as the
 * reference re-plans after a length change (data_thread.rs:1194-1227). */
int thz_tilt_apply(thz_ctx *ctx, size_t npix, const float *d_in, size_t nt_in, const float *d_taper,
                   const int32_t *d_insert_index, size_t nt_out, float *d_out);

/* ------------------------------------------------------------------ */
/* Deconvolution (K12), src/filters/deconvolution.rs + src/filters/psf.rs */
/* ------------------------------------------------------------------ */

/* CubicSplineCoeffs, psf.rs:7-14 (arrays as loaded from psf.npz, io.rs:146-166) */
typedef struct thz_spline {
    const float *knots;    /* n_knots   */
    const float *values;   /* n_knots   */
    const float *coeff_a;  /* n_knots-1 */
    const float *coeff_b;
    const float *coeff_c;
    const float *coeff_d;
    size_t n_knots;
} thz_spline;

/* HybridFit, psf.rs:17-22: a/f + b + spline correction */
typedef struct thz_hybrid_fit {
    float base_a, base_b;
    thz_spline correction;
} thz_hybrid_fit;

/* PSF, psf.rs:201-207 */
typedef struct thz_psf {
    thz_hybrid_fit wx_fit, wy_fit;
    thz_spline x0_spline, y0_spline;
} thz_psf;

/* Deconvolution fields, deconvolution.rs:239-253 (defaults 500 / 25 / 0.1 / 10 / 0.5) */
typedef struct thz_deconv_cfg {
    uint32_t n_iterations;
    uint32_t n_filters;
    float start_freq, end_freq, win_width;
    /* Band-parallel execution across GPUs: this call only sums
     * bands [band_begin, band_end) of the n_filters-band bank; 0, 0 = all bands;
     * band_begin == band_end > 0 = none (more ranks than bands).  The per-rank
     * d_out add up to the full result (all-reduce them) on every path: when a guard
     * or an abort makes the stage pass its input through, the rank that owns band 0
     * writes the input and every other rank zeros, so the all-reduce yields the
     * input once; every rank returns the same status for a guard.  d_img of a band
     * subset is the energy of a partial sum: take the image after the all-reduce.
     * The iteration schedule uses the beam widths of the whole bank. */
    uint32_t band_begin, band_end;
} thz_deconv_cfg;

/* host pieces, exported for tests and for hosts that want to show them:
 * HybridFit::eval_single / eval_single_const_extrap (psf.rs:83-131),
 * create_filter_bank (deconvolution.rs:160-211; filters n_filters x 499),
 * the per-band 2-D PSF (deconvolution.rs:906-960 + psf.rs:228-313; call with
 * out = NULL to get rows/cols). */
int thz_host_psf_eval(const thz_psf *psf, const float *freqs, size_t n, float *wx, float *wy,
                      float *x0, float *y0);
int thz_host_filter_bank(const float *time, size_t nt, const thz_deconv_cfg *cfg, float *filters,
                         float *centers);
int thz_host_band_psf(const thz_psf *psf, float center_freq, float dx, float dy, size_t img_rows,
                      size_t img_cols, float *out, size_t *rows, size_t *cols);

/* Deconvolution::filter, deconvolution.rs:766-1041, on the cube (nx, ny, nt) with
 * nt of the current time axis: FIR bank -> per band energy image ->
 * Richardson–Lucy against the band's Gaussian PSF -> gain sqrt(max(u,0)/d) ->
 * sum over bands of gain x filtered trace; d_img (may be NULL) = sum_t out^2.
 * d_gains_out (may be NULL) receives the (n_filters, nx, ny) gains.
 * abort_flag (may be NULL) is polled between iteration batches like the
 * reference's cancellable loops; *progress (may be NULL) is updated in [0,1].
 * Returns THZ_SKIPPED with out = in when one of the reference's guards applies
 * (empty PSF, image < 16x16, PSF wider than the image), THZ_ERR_ABORTED (out =
 * in) when aborted, THZ_ERR_UNSUPPORTED for traces of more than 7694 samples
 * (the FIR transform's padded length is at most 8192).  Blocking.  The call's device scratch (one padded spectrum
 * per pixel, the bands' images: ~0.6 KB x nt/1000 per pixel and band) and the
 * iteration graphs stay with the context for the next call of the same geometry and is released by a call of
 * another geometry or by thz_destroy.
 * Arithmetic against the reference's: fp32 FIR (the reference's is Complex<f64>); band energies as
 * Parseval's sum minus the two 249-sample edges of the full convolution; every band PSF as two 1-D passes over its
 * profiles — the same sums in another order, also for kernels of <= 256 taps, which the reference sums directly
 * (environment variable THZ_RL_NARROW_EXACT=1: those in the reference's own order, bit for bit).  Cube, gains and image
 * within 1e-5 of the reference's at its defaults (500 iterations, 25 bands). */
int thz_deconvolve(thz_ctx *ctx, const thz_psf *psf, const thz_deconv_cfg *cfg, size_t nx, size_t ny,
                   float dx, float dy, const float *d_in, float *d_out, float *d_img,
                   float *d_gains_out, volatile const int *abort_flag, float *progress);

/* Synthetic input generator for benchmarks and tests (not a reference
 * function): derivative-of-Gaussian pulse + echo + 1 % noise
 * per trace from counter-based Philox4x32-10, identical to tests/synth.py.
 * Trace ids first_trace .. first_trace+ntraces-1; d_time is the nt-sample
 * time axis on the device. */
int thz_synth_cube(thz_ctx *ctx, float *d_out, size_t ntraces, uint64_t first_trace,
                   const float *d_time, uint32_t seed, int subtract_bias);

/* Measurement aid (not a reference function): moves the bytes of thz_pipeline in its access
 * shape — per trace read nt floats, write 2 nf + nf + nf + nt floats to four arrays — with no
 * arithmetic, so that the roofline can be quoted against what the memory system gives this
 * traffic pattern as well as against the 8 TB/s spec.  nt % 8 == 0; time under THZ_STAGE_PROBE. */
int thz_traffic_probe(thz_ctx *ctx, size_t npix, size_t nt, const float *d_in, float *d_fft, float *d_amp,
                      float *d_phase, float *d_data_out);

/* ------------------------------------------------------------------ */
/* Resident cube + whole-chain recompute ("session")                    */
/* ------------------------------------------------------------------ */
/* What the data thread keeps per open file, on the device: the raw cube
 * (after the load-time bias subtraction) and ONE set of outputs — spectrum,
 * amplitudes, phases, final trace cube, image, pixel means — instead of the
 * reference's nine full containers (config.rs:270, main.rs:178-268).  Because
 * the whole default chain is one HBM pass (thz_pipeline), every
 * UpdateType::Filter(idx) (data_thread.rs:1023) is served by re-running it from
 * the raw cube: that is the partial-recompute policy — nothing but the raw cube
 * is cached.  A non-zero tilt changes the trace length and takes the staged
 * path (tilt kernel, re-plan, transforms of the new length). */
typedef struct thz_session thz_session;

typedef struct thz_chain_cfg {
    /* TiltCompensation (tilt_compensation.rs:27-32); active by default with 0/0 */
    int32_t tilt_active;
    double tilt_x_deg, tilt_y_deg;
    /* Time Band Pass before the FFT (band_pass_td_before_fft.rs:28-33, reset :66-72) */
    int32_t td_before_active;
    double td_before_low, td_before_high, td_before_width;
    /* ConfigContainer.fft_window(_type), config.rs:171-213 */
    thz_window_cfg fft_window;
    /* Frequency Band Pass (band_pass_fd.rs:30-37) */
    int32_t fd_active;
    double fd_low, fd_high, fd_width;
    /* Time Band Pass after the inverse FFT (band_pass_td_after_fft.rs) */
    int32_t td_after_active;
    double td_after_low, td_after_high, td_after_width;
    /* pixel means of the ifft stage (math_tools.rs:421-440): 0 none; 1 amplitude / phase sums taken
     * inside the fused launch and avg_fft by linearity from the mean trace (<= 1e-5 of the reference's
     * values, no extra pass over the outputs); 2 the reference's summation order bit for bit (three
     * passes over the outputs; also what 1 falls back to for a tilted cube) */
    int32_t want_means;
    /* ConfigContainer.scale_factor: the chain's first stage, math_tools::scaling
     * (math_tools.rs:242-310).  s > 1 replaces the raw cube by its s x s block
     * means (sum / s^2, also on ragged edges), every later stage and output then
     * lives on the (nx / s, ny / s) grid with dx * s, dy * s (thz_session_grid);
     * s <= 1, or a side shorter than s, leaves the grid alone. */
    int32_t scale_factor;
    /* ConfigContainer.avg_in_fourier_space (config.rs:171-213; default false): the ifft stage then also
     * rebuilds a time trace from the pixel-mean amplitudes and phases — avg_data = C2R(from_polar(avg_signal_fft,
     * avg_phase_fft)) / nt (math_tools.rs:442-470), needs want_means — and every region of interest's roi_data
     * from its own mean amplitudes and phases (:496-529) instead of from the traces (:477-483). */
    int32_t avg_in_fourier_space;
} thz_chain_cfg;

/* Defaults of the reference after OpenFile + reset(): every filter active, bounds
 * = ends of the time axis, widths 2.0 / 0.1 ps, 0.2-5 THz / 0.1 THz, window
 * AdaptedBlackman [1, 7] ps, no tilt. */
int thz_chain_cfg_default(const float *time, size_t nt, thz_chain_cfg *out);

enum {
    THZ_BUF_RAW = 0,        /* (nx, ny, nt) f32 */
    THZ_BUF_FFT = 1,        /* (nx, ny, nf) complex */
    THZ_BUF_AMPLITUDES = 2, /* (nx, ny, nf) */
    THZ_BUF_PHASES = 3,     /* (nx, ny, nf) */
    THZ_BUF_DATA = 4,       /* (nx, ny, nt_out) final trace cube */
    THZ_BUF_IMG = 5,        /* (nx, ny) */
    THZ_BUF_AVG_FFT = 6,    /* (nf) complex   — needs want_means */
    THZ_BUF_AVG_AMPLITUDES = 7, /* (nf) */
    THZ_BUF_AVG_PHASES = 8, /* (nf) */
    THZ_BUF_OPACITY = 9     /* (nx, ny, nt_out) after thz_session_voxels */
};

int thz_session_create(thz_ctx *ctx, size_t nx, size_t ny, size_t nt, const float *time, float dx,
                       float dy, thz_session **out);
void thz_session_destroy(thz_session *s);
/* open_scan_from_thz's in-memory part (io.rs:576-628): H2D, per-trace bias
 * subtraction when asked, intensity image of the raw cube.  `cube` is host
 * memory (nx, ny, nt) C-order, or NULL when the caller fills the raw buffer
 * itself through thz_session_buffer(). */
int thz_session_upload(thz_session *s, const float *cube, int subtract_bias);
/* UpdateType::Filter(1) (ConfigCommand::UpdateFilters, SetDownScaling: data_thread.rs:836-838, 903-905):
 * recomputes every output. */
int thz_session_recompute(thz_session *s, const thz_chain_cfg *cfg);
/* UpdateType::Filter(start_idx), data_thread.rs:1090-1105: the stage walk starts at chain position
 * start_stage of the reference's filter_chain (main.rs:182-247; "initial" is 0):
 *   1 scaling  2 Tilt Compensation  3 Time Band Pass  4 fft  5 Frequency Band Pass (+ the plugins of
 *   thz_session_set_fd_filters)  6 ifft  7 Time Band Pass (after)  8 Deconvolution
 * (UpdateFilter(uuid) sends the filter's own position, :907-921; the fft window commands send fft_index =
 * 3, one in front of the fft stage, :813-836.)  Positions 1-5 run the one-pass chain from the raw cube:
 * the resident spectrum is the band-passed one, so nothing in front of position 6 can restart from an
 * intermediate.  Positions 6 and 7 re-run only C2R -> Time Band Pass -> image on the resident spectrum
 * (a third of the full chain's traffic) — provided the last full recompute used the same settings in
 * front of position 6, else they fall back to the full chain.  Position 8 is thz_session_deconvolve's;
 * here it is a no-op. */
int thz_session_recompute_from(thz_session *s, const thz_chain_cfg *cfg, int start_stage);
/* Further Frequency-domain plugins of the chain (FilterDomain::Frequency, behind "Frequency Band Pass"),
 * as per-bin multipliers for the chain's nf bins (host vectors, copied; NULL removes one): a real one —
 * the water-line notch K14, thz_host_water_line_mask — and a complex one — the reference-pulse Wiener
 * filter K13, thz_host_wiener_filter.  Both ride in the fused launch.  A later recompute whose spectra
 * have another length (tilted cube) returns THZ_ERR_INVALID. */
int thz_session_set_fd_filters(thz_session *s, const float *real_mask, const float *cmask, size_t nf);
/* Regions of interest — ScannedImageFilterData::rois (data_container.rs:109-162), edited by ConfigCommand::AddROI /
 * UpdateROI / DeleteROI (data_thread.rs:922-1000).  Replaces the session's whole set (n_rois = 0 removes it).
 * poly_xy: the polygons' vertices one after the other, (x, y) u64 pairs exactly as the reference keeps them
 * (f64 -> usize by truncation, :936-939: pixels of the RAW grid — average_polygon_roi divides them by the
 * scale factor itself, math_tools.rs:606-609); n_vertices[i] of them belong to region i.  From the next
 * recompute on, the ifft stage's per-region means (math_tools.rs:473-543) and the plot copy-out's
 * (data_thread.rs:1442-1482) are taken with every recompute and read with thz_session_roi. */
int thz_session_set_rois(thz_session *s, size_t n_rois, const size_t *n_vertices, const uint64_t *poly_xy);
size_t thz_session_roi_count(const thz_session *s);
/* One region's vectors after a recompute; host pointers, any may be NULL.  The masks follow the reference's
 * integer rule bit for bit (thz_roi_mask); the sampled pixel of mask position (x, y) is [shape0 - y - 1, x]
 * (math_tools.rs:647).  With want_means == 2 the pixels are added in the reference's order — y outer, x inner,
 * sequential f32 — and divided by the count: the means are those of the resident arrays bit for bit; otherwise
 * (0, 1) the sums are taken in parallel (<= 2e-6 of the largest value away). */
typedef struct thz_roi_out {
    float *signal_fft; /* nf_out  roi_signal_fft: mean amplitudes (:485, data_thread.rs:1453-1461) */
    float *phase_fft;  /* nf_out  roi_phase_fft: mean unwrapped phases (:486, :1464-1472) */
    float *signal;     /* nt_out  data.roi_signal of the plot copy-out: mean of the chain's FINAL traces
                                   (data_thread.rs:1445-1451) or, with avg_in_fourier_space, roi_data (:1476-1482) */
    float *roi_data;   /* nt_out  the ifft stage's roi_data: mean of the stage's input traces — the fft stage's
                                   windowed `data` — (math_tools.rs:477-483) or, with avg_in_fourier_space,
                                   C2R(from_polar(signal_fft, phase_fft), Im X[0] := 0) / nt (:496-529; where realfft
                                   would refuse the spectrum — Im of the last bin of an even length != 0 — the
                                   reference's fallback: the mean of the traces, :530-538) */
    uint32_t *count;   /* pixels inside the polygon */
} thz_roi_out;
int thz_session_roi(thz_session *s, size_t roi, const thz_roi_out *out);

/* Grid of the last recompute's outputs (the raw grid until then): every buffer
 * except THZ_BUF_RAW has nx * ny pixels of this grid.  Any pointer may be NULL. */
int thz_session_grid(const thz_session *s, size_t *nx, size_t *ny, float *dx, float *dy);
/* UpdateType::Filter(<Deconvolution>): the chain's last stage (FilterDomain::
 * TimeAfterFFTPrioLast, deconvolution.rs:751) — Deconvolution::filter
 * (deconvolution.rs:766-1041) on the "Time Band Pass" output of the last
 * recompute, with the session's dx / dy.  Its result is THZ_BUF_DATA /
 * THZ_BUF_IMG (and what thz_session_voxels / thz_session_plot read) until the
 * next thz_session_recompute: the reference does not re-run the deconvolution
 * when another filter is updated (data_thread.rs:1080, 1139-1149), the stage
 * then passes its input through (:1186-1188).  Calling it again deconvolves the
 * Time Band Pass output again, not the previous result.  Returns THZ_OK,
 * THZ_SKIPPED (a guard returned the input unchanged; the output is that copy)
 * or a negative code (THZ_ERR_ABORTED: the final output is the input again). */
int thz_session_deconvolve(thz_session *s, const thz_psf *psf, const thz_deconv_cfg *cfg,
                           volatile const int *abort_flag, float *progress);
/* trace length of the final cube (nt, or nt + 2*steps after a tilt) and its axis */
size_t thz_session_nt_out(const thz_session *s);
int thz_session_time_out(const thz_session *s, float *time /* nt_out */);
/* device pointer of a resident buffer; NULL if absent — every output buffer (everything but THZ_BUF_RAW
 * and THZ_BUF_IMG) is absent until a recompute has run after the latest upload */
void *thz_session_buffer(thz_session *s, int which);
/* copies pixels [pix0, pix0+npix) of a per-pixel buffer (or the whole vector for
 * the AVG_* ones, pix0 = 0, npix = 1) to the host: the selected-pixel trace, a
 * tile, or everything.  THZ_ERR_NOT_READY for an absent buffer, THZ_ERR_INVALID for a
 * pixel range beyond the grid of the recompute that filled it. */
int thz_session_download(thz_session *s, int which, size_t pix0, size_t npix, void *dst);

/* ------------------------------------------------------------------ */
/* Multi-GPU: x-slab tiles of one cube over the GPUs of a node          */
/* ------------------------------------------------------------------ */
/* Every (x, y) trace is independent, so a cube is split into contiguous slabs of x rows,
 * one per GPU — the split rayon makes over Axis(0) in the reference's pixel loops (math_tools.rs:333-339,
 * 545-568).  The data path needs no collective; per recompute there are two exchange steps, both RCCL calls
 * made by this library on the members' own streams (no host sync between kernels and collectives):
 *   C2  ncclAllReduce(sum) of the slabs' pixel-sum vectors (2 nf floats) -> pixel means on every member
 *   C1  grouped ncclSend / ncclRecv gather of per-pixel results to rank 0: the image always, the final
 *       trace cube / every output on request (thz_gather)
 * A group is either ONE process driving n devices — the shape of the reference, whose single data thread
 * (data_thread.rs:162-174) would own all of them — or one member of a one-process-per-GPU launch
 * (torchrun-style); the calls below are the same in both.  librccl is opened on first use of a group with
 * more than one device, not at library load. */
typedef struct thz_group thz_group;
#define THZ_GROUP_ID_BYTES 128 /* = NCCL_UNIQUE_ID_BYTES */

/* rows [*x0, *x0 + *n) of `rank`: nx / world rows each, the remainder spread over the first ranks */
int thz_host_slab(size_t nx, int world, int rank, size_t *x0, size_t *n);

/* one process, n devices (ncclCommInitAll).  Members that all name the SAME device form a group without a
 * fabric — collectives become device-local copies on that device — which is how a one-GPU box exercises
 * the slab logic; a mix of repeated and distinct devices is THZ_ERR_INVALID. */
int thz_group_create(const int *devices, int n, thz_group **out);
/* one process per GPU: rank 0 calls thz_group_unique_id and ships the THZ_GROUP_ID_BYTES to every other
 * rank by any means (a file, MPI, torch.distributed ...); every rank then calls thz_group_create_rank
 * (ncclCommInitRank; collective: returns when all `world` ranks have called it). */
int thz_group_unique_id(void *id);
int thz_group_create_rank(int device, int rank, int world, const void *id, thz_group **out);
void thz_group_destroy(thz_group *g);
const char *thz_group_last_error(const thz_group *g);
int thz_group_world(const thz_group *g);        /* ranks in the group */
int thz_group_local_count(const thz_group *g);  /* members this process drives */
int thz_group_rank(const thz_group *g, int i);  /* rank of local member i */
thz_ctx *thz_group_ctx(thz_group *g, int i);    /* context of local member i (owned by the group) */
/* C2 / C1 as stand-alone calls, for callers that drive the stage entry points themselves: d_bufs / d_send
 * hold one device pointer per LOCAL member (on that member's device); counts has one entry per RANK;
 * d_recv_root (on rank 0's device; ignored in processes that do not drive rank 0) receives the rows in
 * rank order.  Enqueued on the members' streams. */
int thz_group_all_reduce_sum(thz_group *g, float *const *d_bufs, size_t count);
int thz_group_all_reduce_u64(thz_group *g, uint64_t *const *d_bufs, size_t count);
int thz_group_gather(thz_group *g, const float *const *d_send, const size_t *counts, float *d_recv_root);
int thz_group_sync(thz_group *g); /* waits for every local member's stream */

/* what C1 brings to rank 0 (gathering everything is xGMI-bound — 6 GiB per peer at
 * 1024 x 1024 x 4096 against 14 ms of compute — the GUI reads only the small products) */
typedef enum thz_gather {
    THZ_GATHER_SMALL = 0, /* image + pixel means (what img_lock / PlotDataContainer readers need) */
    THZ_GATHER_TIME = 1,  /* + the final trace cube (the 3-D tab's input) */
    THZ_GATHER_ALL = 2    /* + spectrum, amplitudes, phases: a complete ScannedImageFilterData on rank 0 */
} thz_gather;

/* A session over a group: member i keeps slab i of the raw cube and of every output resident.  Everything a session
 * does shards (round 3): a non-zero tilt (the plan is made for the whole grid), scale_factor > 1 (a block of s x s
 * pixels whose rows lie in two slabs belongs to the slab that holds its LAST row: the slab in front hands over the
 * partial sums of its rows, the sum continues in the reference's order — THZ_ERR_UNSUPPORTED only when a slab has
 * fewer rows than the scale factor), want_means 2 (the reference's sequential means, slab after slab on a carried
 * running sum: bit for bit one session's, serial by construction). */
typedef struct thz_group_session thz_group_session;
int thz_group_session_create(thz_group *g, size_t nx, size_t ny, size_t nt, const float *time, float dx, float dy,
                             thz_group_session **out);
void thz_group_session_destroy(thz_group_session *gs);
/* the slab session of local member i (its buffers, plot copy-out, voxels ...); rows via thz_host_slab */
thz_session *thz_group_session_member(thz_group_session *gs, int i);
/* `cube`: the WHOLE (nx, ny, nt) host cube — each member uploads its rows — or NULL when the caller filled
 * the members' THZ_BUF_RAW itself.  Collective (C2 of the raw pixel sums). */
int thz_group_session_upload(thz_group_session *gs, const float *cube, int subtract_bias);
/* UpdateType::Filter(start_stage) on every slab + C2 + C1.  Collective; blocking. */
int thz_group_session_recompute(thz_group_session *gs, const thz_chain_cfg *cfg, int start_stage, int gather);
/* UpdateType::Filter(<Deconvolution>) over the group — BASELINE config 4's "1 -> 2 GPUs".  The stage's transform, band
 * energies and recombination are per PIXEL, its Richardson-Lucy iterations per BAND over the whole image: every member
 * runs the per-pixel parts for its own rows and the iterations for its own bands (contiguous ranges dealt out by a
 * fitted time model), and what crosses the fabric is two sets of 2-D images — n_filters x Nx x Ny energies out, as many
 * gains back (grouped ncclBroadcast) — not the cube (round 2's form gathered and all-reduced it).  The output slab stays
 * on its member; image (and the final cube, if the last recompute gathered it) are re-gathered to rank 0.  Members of
 * one process run on host threads.  A guard of the reference (the same on every rank) or an abort / error on any rank
 * (agreed by a one-float all-reduce after each part) makes every slab keep its input.  Same return codes as
 * thz_session_deconvolve.  DESIGN.md §5: expected times on 2 / 4 / 8 GPUs from measured phase times. */
int thz_group_session_deconvolve(thz_group_session *gs, const thz_psf *psf, const thz_deconv_cfg *cfg,
                                 volatile const int *abort_flag, float *progress);
/* Regions of interest over the whole (nx, ny) grid (thz_session_set_rois): every slab sums its rows of the
 * whole grid's mask — the sampled row index shape0 - y - 1 (math_tools.rs:647) runs along the sharded axis —,
 * the recompute's C2 all-reduces the regions' sums with the pixel sums, and every member divides by the grid's
 * pixel count.  thz_group_session_roi reads local member 0's copy (identical on all).  Collective like the
 * recompute itself: every rank sets the same regions. */
int thz_group_session_set_rois(thz_group_session *gs, size_t n_rois, const size_t *n_vertices, const uint64_t *poly_xy);
int thz_group_session_roi(thz_group_session *gs, size_t roi, const thz_roi_out *out);
/* gathered results on rank 0's device after a recompute: THZ_BUF_IMG (nx, ny) always; THZ_BUF_DATA with
 * THZ_GATHER_TIME / ALL; THZ_BUF_FFT / AMPLITUDES / PHASES with ALL; THZ_BUF_AVG_* (on every member these
 * are also in its slab session).  NULL when absent or when this process does not drive rank 0. */
void *thz_group_session_result(thz_group_session *gs, int which);
/* the whole grid of the last recompute's outputs (thz_session_grid over all slabs): (nx, ny) of the raw grid, or
 * behind a scaling stage (nx / s, ny / s); what the gathered buffers and thz_group_session_download index */
int thz_group_session_grid(const thz_group_session *gs, size_t *nx, size_t *ny);
int thz_group_session_download(thz_group_session *gs, int which, size_t pix0, size_t npix, void *dst);

/* Plot copy-out of UpdateType::Plot (data_thread.rs:1337-1432): everything the
 * right-hand panel plots for the selected pixel, in one call.  Host pointers, any
 * may be NULL.  px, py index the (nx, ny) grid (pixel_selected / scaling). */
typedef struct thz_plot_out {
    float *signal;              /* nt      raw trace, filter_data.first()               :1344-1362 */
    float *signal_fft;          /* nf_out  amplitudes right after the fft stage         :1366-1380 */
    float *phase_fft;           /* nf_out  phases right after the fft stage                        */
    float *filtered_signal;     /* nt_out  final trace, filter_data.last()              :1384-1396 */
    float *filtered_signal_fft; /* nf_out  band-passed amplitudes                       :1398-1408 */
    float *filtered_phase_fft;  /* nf_out                                               :1409-1419 */
    float *avg_signal;          /* nt_out  pixel mean of the final cube, or — avg_in_fourier_space of the last
                                            recompute — the ifft stage's avg_data            :1422-1432 */
    float *avg_signal_fft;      /* nf_out  pixel mean of the amplitudes (needs want_means) :1434   */
    float *avg_phase_fft;       /* nf_out                                                  :1435   */
} thz_plot_out;
int thz_session_plot(thz_session *s, size_t px, size_t py, const thz_plot_out *out);

/* ConfigCommand::OpenRef (data_thread.rs:372-588): a reference pulse read from its own file
 * becomes a length-nt vector on the scan's time axis — zero-padded index shift by
 * round((scan_time[0] - ref_time[0]) / ref_dt) (:405-481) — is windowed with the *reference
 * file's* time axis (:490-515) and transformed with the scan's plan (:517-533).
 * thz_host_align_reference returns 0 (lengths and origin already match), 1 (shifted) or
 * 2 (naive pad / truncate for degenerate axes). */
int thz_host_align_reference(const float *scan_time, size_t nt, const float *ref_time, const float *ref_signal,
                             size_t nref, float *out /* nt */);
/* all host pointers; reference_out (nt) = aligned and windowed pulse (roi_data), amplitudes /
 * phases (nf) = |X| and numpy_unwrap(arg X) (roi_signal_fft / roi_phase_fft).  Plans for
 * scan_time if the context's axis differs.  THZ_ERR_INVALID where the reference panics: a
 * non-adapted window with nref != nt. */
int thz_reference_spectrum(thz_ctx *ctx, const float *scan_time, size_t nt, const float *ref_time,
                           const float *ref_signal, size_t nref, const thz_window_cfg *window,
                           float *reference_out, float *amplitudes, float *phases);

/* calculate_optical_properties (math_tools.rs:663-701): refractive index, absorption
 * coefficient and extinction coefficient per bin from sample and reference amplitude /
 * phase spectra (host vectors; f32 in the reference's operation order). */
int thz_host_optical_properties(const float *sample_amp, const float *sample_phase, const float *ref_amp,
                                const float *ref_phase, const float *freq, size_t nf, float thickness,
                                float *refractive_index, float *absorption_coeff, float *extinction_coeff);

/* ------------------------------------------------------------------ */
/* 3-D voxel envelope (gui/threed_plot.rs:80-276)                       */
/* ------------------------------------------------------------------ */
/* The data thread rebuilds the voxel instances of the 3-D tab after every
 * recompute (update_intensity_image, data_thread.rs:48-101): per trace
 * (v^2)^contrast -> 1-D Gaussian -> max / min rule; then the max_instances-th
 * largest opacity of the whole cube (select_nth_unstable_by, :205-214) becomes
 * the effective threshold and every voxel >= it becomes one instance. */
typedef struct thz_voxel_cfg {
    float opacity_threshold; /* gui_settings.opacity_threshold, application.rs:202 (0.1) */
    float contrast;          /* contrast_3d (2.0) */
    float sigma;             /* kernel_sigma (3.0) */
    int32_t radius;          /* kernel_radius (9); slider 1..50 */
} thz_voxel_cfg;

/* bevy_voxel_plot::InstanceData as filled at threed_plot.rs:260-264 */
typedef struct thz_voxel_instance {
    float position[3];
    float scale;
    float color[4]; /* linear RGB of the jet colour, alpha = opacity */
} thz_voxel_instance;

#define THZ_VOXEL_MAX_INSTANCES 2000000u /* threed_plot.rs:206 */

int thz_voxel_cfg_default(thz_voxel_cfg *out);
/* gaussian_kernel1d (:80-101); out has 2*radius+1 entries */
int thz_host_gaussian_kernel1d(float sigma, int radius, float *out);
/* d_data (npix, nt) -> d_opacity (npix, nt), nt <= 8192; the two may not alias */
int thz_voxel_opacity(thz_ctx *ctx, size_t npix, size_t nt, const float *d_data, const thz_voxel_cfg *cfg,
                      float *d_opacity);
/* Radix select, one level: d_hist (2048 u64, device) += histogram of the level's
 * bits among the n values whose higher bits equal `prefix` (level 0: 11 bits of
 * all values; level 1: next 11; level 2: last 10).  At level 0 `prefix` is a floor
 * bin instead: keys of lower bins are counted in it (0 = full histogram); if the
 * walk ends in a non-zero floor bin, level 0 is repeated with floor 0.
 * Histograms of the tiles of a cube add up, so ranks all-reduce d_hist between
 * levels. */
int thz_select_histogram(thz_ctx *ctx, const float *d_vals, size_t n, int level, uint32_t prefix,
                         uint64_t *d_hist);
/* host walk of one level: the bin holding the k-th largest (k >= 1) and its rank
 * inside the bin; nbins = 2048 (levels 0, 1) or 1024 (level 2) */
int thz_host_select_step(const uint64_t *hist, int nbins, uint64_t k, int *bin, uint64_t *k_rem);
/* value from the three bins of the levels */
float thz_host_select_value(int bin0, int bin1, int bin2);
/* the three levels on one GPU: k-th largest of n values (1 <= k <= n) */
int thz_kth_largest(thz_ctx *ctx, const float *d_vals, size_t n, uint64_t k, float *out);
/* effective threshold (:205-214): 0.0 when n <= max_instances */
int thz_voxel_threshold(thz_ctx *ctx, const float *d_opacity, size_t n, uint64_t max_instances, float *out);
/* Instance loop (:216-271) over a (gw, gh, gd) opacity cube, in x, y, z order.
 * x0 / gw_total place a tile of x-rows inside the whole grid (x0 = 0, gw_total =
 * gw for one GPU).  Writes at most `capacity` records to d_out and the number of
 * voxels >= threshold to *count; cube_dims = {cube_width, cube_height, cube_depth}. */
int thz_voxel_instances(thz_ctx *ctx, const float *d_opacity, size_t gw, size_t gh, size_t gd, size_t x0,
                        size_t gw_total, float threshold, float time_span, int scaling, size_t orig_w,
                        size_t orig_h, size_t orig_d, thz_voxel_instance *d_out, uint64_t capacity,
                        uint64_t *count, float *cube_dims);

/* update_intensity_image's 3-D part for a session (data_thread.rs:82-101): opacity cube of the
 * session's final trace cube, effective threshold, instance list — downloaded to `host_out`
 * (capacity records; NULL with capacity 0 to only count).  time_span = last - first sample of the
 * final time axis; scaling / orig_* as in thz_voxel_instances.  The opacity cube stays resident in
 * the session (THZ_BUF_OPACITY) until the next call. */
int thz_session_voxels(thz_session *s, const thz_voxel_cfg *cfg, uint64_t max_instances, int scaling,
                       size_t orig_w, size_t orig_h, size_t orig_d, thz_voxel_instance *host_out,
                       uint64_t capacity, uint64_t *count, float *threshold, float *cube_dims);

/* Per-stage device time of the most recent call of each kind, the value the
 * reference shows next to each filter (filter.rs:607-621).  `stage` is one
 * of the THZ_STAGE_* ids. */
enum {
    THZ_STAGE_FFT = 0,
    THZ_STAGE_FD_MASK = 1,
    THZ_STAGE_IFFT = 2,
    THZ_STAGE_PIPELINE = 3,
    THZ_STAGE_TD_WINDOW = 4,
    THZ_STAGE_INTENSITY = 5,
    THZ_STAGE_MEAN = 6,
    THZ_STAGE_ROI = 7,
    THZ_STAGE_VOXEL_OPACITY = 8,
    THZ_STAGE_VOXEL_SELECT = 9,
    THZ_STAGE_VOXEL_EMIT = 10,
    THZ_STAGE_PROBE = 11,
    THZ_STAGE_COUNT = 12
};
/* hipEvent bracketing of every stage call on the context's stream.
 *   0  off (default)
 *   1  immediate: the call waits for its kernel; thz_stage_time_ns() then
 *      returns the device time of the latest call of that stage
 *   2  deferred: events are recorded without any host wait;
 *      thz_timing_collect() later synchronises once and returns the summed
 *      device time and the number of calls of `stage` since the last collect */
int thz_enable_timing(thz_ctx *ctx, int mode);
int thz_stage_time_ns(thz_ctx *ctx, int stage, uint64_t *ns);
int thz_timing_collect(thz_ctx *ctx, int stage, uint64_t *total_ns, uint64_t *count);

/* Kernel variant actually used for the current nt ("stockham-lds-r2", …),
 * for logs and for the tests that assert the native path ran. */
const char *thz_kernel_variant(const thz_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* THZGPU_H */
