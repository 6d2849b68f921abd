/* thz_oracle.c — CPU restatement of the reference's data_thread recompute path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (thz_image_explorer_amd/,
 * include/, libthzgpu.so) may link, import or call this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker and as the reported CPU baseline ("port").
 *
 * PARITY STATUS: the reference (Rust) cannot be compiled here and ships no
 * numeric golden vectors for this path.  This restatement
 * is pinned by (i) the reference's own unit-test properties, re-run against it
 * in tests/test_oracle_reference_properties.py, and (ii) numpy-fp64 golden
 * vectors in tests/golden/ for the FFT arithmetic that lives in the un-vendored
 * realfft/rustfft crates.  Beyond those properties parity is "unpinned".
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * it restates.  Arithmetic is fp32 in the reference's operation order unless
 * a function says otherwise.  Build: see oracle/Makefile (-ffp-contract=off so
 * that no FMA is formed where the Rust code has separate mul and add).
 */
#define _GNU_SOURCE
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define T float
#define SUF _f
#include "fft_generic.inc"
#undef T
#undef SUF

#define T double
#define SUF _d
#include "fft_generic.inc"
#undef T
#undef SUF

#define PI_F 3.14159274101257324219f /* std::f32::consts::PI */

/* ---------------------------------------------------------------------------
 * Windows — src/math_tools.rs:81-198
 * ------------------------------------------------------------------------- */

/* math_tools.rs:81-90 */
float thz_oracle_blackman_window(float n, float m)
{
    float res = 0.42f - 0.5f * cosf(2.0f * PI_F * n / m) + 0.08f * cosf(4.0f * PI_F * n / m);
    if (isnan(res)) return 1.0f;
    if (res < 0.0f) return 0.0f;
    if (res > 1.0f) return 1.0f;
    return res;
}

/* math_tools.rs:102-122.  `time` has `len` entries and is the slice the
 * caller passes (its own first/last element are the taper origins, a'-2). */
void thz_oracle_apply_adapted_blackman(float *signal, const float *time, int len, float lower_bound,
                                       float upper_bound)
{
    if (len <= 0) return;
    float t0 = time[0], tn = time[len - 1];
    for (int i = 0; i < len; ++i) {
        float t = time[i];
        if (t <= lower_bound + t0) {
            signal[i] *= thz_oracle_blackman_window(t - t0, 2.0f * lower_bound);
        } else if (t >= tn - upper_bound) {
            signal[i] *= thz_oracle_blackman_window(t - (tn - upper_bound * 2.0f), 2.0f * upper_bound);
        }
    }
}

/* math_tools.rs:131-135 */
static void normalize_time(const float *time, int len, float *out)
{
    float mn = INFINITY, mx = -INFINITY;
    for (int i = 0; i < len; ++i) { mn = fminf(mn, time[i]); mx = fmaxf(mx, time[i]); }
    for (int i = 0; i < len; ++i) out[i] = (time[i] - mn) / (mx - mn);
}

/* window type ids follow the order of FftWindowType, math_tools.rs:35-46 */
enum { WIN_ADAPTED_BLACKMAN = 0, WIN_BLACKMAN = 1, WIN_HANNING = 2, WIN_HAMMING = 3, WIN_FLAT_TOP = 4 };

/* math_tools.rs:145-198 (+ dispatch at 356-371) */
void thz_oracle_apply_window(int type, float *signal, const float *time, int len, float lo, float hi)
{
    if (type == WIN_ADAPTED_BLACKMAN) {
        thz_oracle_apply_adapted_blackman(signal, time, len, lo, hi);
        return;
    }
    float *tau = (float *)malloc(sizeof(float) * (size_t)(len > 0 ? len : 1));
    normalize_time(time, len, tau);
    for (int i = 0; i < len; ++i) {
        float t = tau[i];
        switch (type) {
        case WIN_HAMMING: signal[i] *= 0.54f - 0.46f * cosf(2.0f * PI_F * t); break;
        case WIN_HANNING: signal[i] *= 0.5f * (1.0f - cosf(2.0f * PI_F * t)); break;
        case WIN_BLACKMAN:
            signal[i] *= 0.42f - 0.5f * cosf(2.0f * PI_F * t) + 0.08f * cosf(4.0f * PI_F * t);
            break;
        case WIN_FLAT_TOP:
            signal[i] *= 1.0f - 1.93f * cosf(2.0f * PI_F * t) + 1.29f * cosf(4.0f * PI_F * t)
                         - 0.388f * cosf(6.0f * PI_F * t) + 0.028f * cosf(8.0f * PI_F * t);
            break;
        default: break;
        }
    }
    free(tau);
}

/* ---------------------------------------------------------------------------
 * numpy_unwrap — src/math_tools.rs:211-240 (period given)
 * ------------------------------------------------------------------------- */
void thz_oracle_numpy_unwrap(const float *x, int n, float period, float *out)
{
    if (n <= 0) return;
    float prev_val = x[0], prev_unwrapped = x[0];
    out[0] = x[0];
    for (int i = 1; i < n; ++i) {
        float val = x[i];
        float diff = val - prev_val;
        if (diff > period / 2.0f) diff -= period;
        else if (diff < -period / 2.0f) diff += period;
        float u = prev_unwrapped + diff;
        prev_val = val;
        prev_unwrapped = u;
        out[i] = u;
    }
}

/* frequency axis — src/io.rs:614-621, data_thread.rs:1197-1207 (a16) */
void thz_oracle_frequency_axis(const float *time, int nt, float *freq)
{
    float rng = time[nt - 1] - time[0];
    int nf = nt / 2 + 1;
    for (int i = 0; i < nf; ++i) freq[i] = (float)i / rng;
}

/* ---------------------------------------------------------------------------
 * Raw transforms (realfft stand-ins), exported for the golden-vector tests
 * ------------------------------------------------------------------------- */
void thz_oracle_rfft_f32(const float *x, int n, float *out_interleaved)
{
    rplan_f *r = rplan_new_f(n);
    cpx_f *work = (cpx_f *)malloc(sizeof(cpx_f) * (size_t)(2 * n + 4));
    rfft_f(r, x, (cpx_f *)out_interleaved, work);
    free(work);
    rplan_free_f(r);
}

void thz_oracle_irfft_f32(const float *X_interleaved, int n, float *x)
{
    rplan_f *r = rplan_new_f(n);
    cpx_f *work = (cpx_f *)malloc(sizeof(cpx_f) * (size_t)(2 * n + 4));
    irfft_f(r, (const cpx_f *)X_interleaved, x, work);
    free(work);
    rplan_free_f(r);
}

void thz_oracle_rfft_f64(const double *x, int n, double *out_interleaved)
{
    rplan_d *r = rplan_new_d(n);
    cpx_d *work = (cpx_d *)malloc(sizeof(cpx_d) * (size_t)(2 * n + 4));
    rfft_d(r, x, (cpx_d *)out_interleaved, work);
    free(work);
    rplan_free_d(r);
}

/* O(n^2) fp64 DFT with exact-argument twiddles: the independent truth the
 * mixed-radix code above is itself tested against. */
void thz_oracle_rdft_direct_f64(const double *x, int n, double *out_interleaved)
{
    int nf = n / 2 + 1;
    for (int k = 0; k < nf; ++k) {
        long double sr = 0, si = 0;
        for (int t = 0; t < n; ++t) {
            long idx = ((long)k * t) % n;
            long double a = -2.0L * 3.141592653589793238462643383279502884L * (long double)idx / (long double)n;
            sr += x[t] * cosl(a);
            si += x[t] * sinl(a);
        }
        out_interleaved[2 * k] = (double)sr;
        out_interleaved[2 * k + 1] = (double)si;
    }
}

/* ---------------------------------------------------------------------------
 * Stage: fft — src/math_tools.rs:330-398
 *   data (nx,ny,nt) is windowed IN PLACE (the stage's `data` output is the
 *   windowed trace, :356-371); fft/amplitudes/phases (nx,ny,nf) are written.
 *   Parallel over Axis(0) like into_par_iter at :333-339.
 * ------------------------------------------------------------------------- */
void thz_oracle_fft_stage(float *data, const float *time, int nx, int ny, int nt, int window_type,
                          float win_lo, float win_hi, float *fft_interleaved, float *amplitudes,
                          float *phases, int nthreads)
{
    int nf = nt / 2 + 1;
    rplan_f *r = rplan_new_f(nt);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
#endif
    {
        cpx_f *work = (cpx_f *)malloc(sizeof(cpx_f) * (size_t)(2 * nt + 4));
        cpx_f *spec = (cpx_f *)malloc(sizeof(cpx_f) * (size_t)nf);
        float *ph = (float *)malloc(sizeof(float) * (size_t)nf);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int x = 0; x < nx; ++x) {
            for (int y = 0; y < ny; ++y) {
                size_t p = (size_t)x * ny + y;
                float *tr = data + p * nt;
                thz_oracle_apply_window(window_type, tr, time, nt, win_lo, win_hi);
                rfft_f(r, tr, spec, work);
                memcpy(fft_interleaved + p * nf * 2, spec, sizeof(cpx_f) * (size_t)nf);
                float *amp = amplitudes + p * nf;
                for (int k = 0; k < nf; ++k) {
                    amp[k] = hypotf(spec[k].re, spec[k].im); /* Complex::norm, :384 */
                    ph[k] = atan2f(spec[k].im, spec[k].re);  /* Complex::arg,  :387 */
                }
                thz_oracle_numpy_unwrap(ph, nf, 2.0f * PI_F, phases + p * nf);
            }
        }
        free(work); free(spec); free(ph);
    }
    rplan_free_f(r);
}

/* ---------------------------------------------------------------------------
 * Frequency band pass — src/filters/band_pass_fd.rs:122-220
 *   Index rule (:135-152) and taper (:162-168) are computed once; the per
 *   pixel multiply (:184-187) and zero padding (:194-212) follow.
 *   window_out (nf floats, may be NULL) receives the full-length multiplier
 *   (0 outside [lower,upper)) — the vector the GPU kernels consume.
 * ------------------------------------------------------------------------- */
void thz_oracle_fd_bandpass_indices(const float *frequency, int nf, double low, double high,
                                    int *lower_out, int *upper_out)
{
    float safe_low = (float)(low > 0.0 ? low : 0.0);
    double last = nf > 0 ? (double)frequency[nf - 1] : 10.0;
    float safe_high = (float)(high < last ? high : last);
    int lower = 0, upper = nf;
    for (int i = 0; i < nf; ++i) if (frequency[i] >= safe_low) { lower = i; break; }
    /* position() -> unwrap_or(0) when nothing matches */
    {
        int found = 0;
        for (int i = 0; i < nf; ++i) if (frequency[i] >= safe_low) { found = 1; break; }
        if (!found) lower = 0;
    }
    {
        int found = 0;
        for (int i = nf - 1; i >= 0; --i) if (frequency[i] <= safe_high) { upper = i + 1; found = 1; break; }
        if (!found) upper = nf;
    }
    *lower_out = lower;
    *upper_out = upper;
}

void thz_oracle_fd_bandpass_window(const float *frequency, int nf, double low, double high,
                                   double window_width, float *window_out, int *lower_out,
                                   int *upper_out)
{
    int lower, upper;
    thz_oracle_fd_bandpass_indices(frequency, nf, low, high, &lower, &upper);
    for (int i = 0; i < nf; ++i) window_out[i] = 0.0f;
    if (upper > lower) {
        for (int i = lower; i < upper; ++i) window_out[i] = 1.0f;
        thz_oracle_apply_adapted_blackman(window_out + lower, frequency + lower, upper - lower,
                                          (float)window_width, (float)window_width);
    }
    if (lower_out) *lower_out = lower;
    if (upper_out) *upper_out = upper;
}

void thz_oracle_fd_bandpass(float *fft_interleaved, float *amplitudes, const float *frequency,
                            size_t npix, int nf, double low, double high, double window_width)
{
    float *w = (float *)malloc(sizeof(float) * (size_t)nf);
    int lower, upper;
    thz_oracle_fd_bandpass_window(frequency, nf, low, high, window_width, w, &lower, &upper);
    for (size_t p = 0; p < npix; ++p) {
        float *f = fft_interleaved + p * nf * 2;
        float *a = amplitudes + p * nf;
        for (int k = 0; k < nf; ++k) {
            if (k >= lower && k < upper) {
                f[2 * k] = f[2 * k] * w[k];
                f[2 * k + 1] = f[2 * k + 1] * w[k];
                a[k] = a[k] * w[k];
            } else {
                f[2 * k] = 0.0f; f[2 * k + 1] = 0.0f; a[k] = 0.0f;
            }
        }
    }
    free(w);
}

/* ---------------------------------------------------------------------------
 * Time band pass — src/filters/band_pass_td_before_fft.rs:124-182 (and the
 * identical _after_fft.rs).  low/high are in/out: the filter clamps its own
 * fields (:137-138).
 * ------------------------------------------------------------------------- */
void thz_oracle_td_bandpass_indices(const float *time, int nt, double *low, double *high,
                                    int *lower_out, int *upper_out)
{
    float min_time = nt > 0 ? time[0] : 0.0f, max_time = nt > 0 ? time[nt - 1] : 0.0f;
    if (*low < (double)min_time) *low = (double)min_time;
    if (*high > (double)max_time) *high = (double)max_time;
    int lower = 0;
    for (int i = 0; i < nt; ++i) if (time[i] >= (float)*low) { lower = i; break; }
    int upper = nt > 0 ? nt - 1 : 0; /* unwrap_or_else(len.saturating_sub(1)) */
    for (int i = 0; i < nt; ++i) if (time[i] >= (float)*high) { upper = i; break; }
    if (upper < lower + 1) upper = lower + 1;
    if (upper > nt) upper = nt;
    *lower_out = lower;
    *upper_out = upper;
}

/* full-length multiplier vector (0 outside [lower,upper)) for the GPU side */
void thz_oracle_td_bandpass_window(const float *time, int nt, double *low, double *high,
                                   double window_width, float *window_out, int *lower_out,
                                   int *upper_out)
{
    int lower, upper;
    thz_oracle_td_bandpass_indices(time, nt, low, high, &lower, &upper);
    for (int i = 0; i < nt; ++i) window_out[i] = 0.0f;
    for (int i = lower; i < upper; ++i) window_out[i] = 1.0f;
    thz_oracle_apply_adapted_blackman(window_out + lower, time + lower, upper - lower,
                                      (float)window_width, (float)window_width);
    if (lower_out) *lower_out = lower;
    if (upper_out) *upper_out = upper;
}

void thz_oracle_td_bandpass(float *data, const float *time, size_t npix, int nt, double *low,
                            double *high, double window_width)
{
    int lower, upper;
    thz_oracle_td_bandpass_indices(time, nt, low, high, &lower, &upper);
    for (size_t p = 0; p < npix; ++p) {
        float *tr = data + p * nt;
        for (int i = 0; i < lower; ++i) tr[i] = 0.0f;
        for (int i = upper; i < nt; ++i) tr[i] = 0.0f;
        thz_oracle_apply_adapted_blackman(tr + lower, time + lower, upper - lower,
                                          (float)window_width, (float)window_width);
    }
}

/* ---------------------------------------------------------------------------
 * Intensity image — src/data_thread.rs:1288-1307, src/io.rs:588-594
 * ------------------------------------------------------------------------- */
void thz_oracle_intensity(const float *data, size_t npix, int nt, float *img)
{
    for (size_t p = 0; p < npix; ++p) {
        const float *tr = data + p * nt;
        float s = 0.0f;
        for (int i = 0; i < nt; ++i) s += tr[i] * tr[i];
        img[p] = s;
    }
}

/* load-time bias subtraction — src/io.rs:578-586 */
void thz_oracle_subtract_bias(float *data, size_t npix, int nt)
{
    for (size_t p = 0; p < npix; ++p) {
        float *tr = data + p * nt;
        float off = tr[0];
        for (int i = 0; i < nt; ++i) tr[i] = tr[i] - off;
    }
}

/* ---------------------------------------------------------------------------
 * Pixel means — src/math_tools.rs:421-440: mean_axis(0) then mean_axis(0),
 * i.e. (1/ny) * sum_y [ (1/nx) * sum_x v[x,y,:] ], sequential fp32.
 * `ncomp` = 1 for real arrays, 2 for interleaved complex.
 * ------------------------------------------------------------------------- */
void thz_oracle_pixel_mean(const float *arr, int nx, int ny, int len, int ncomp, float *out)
{
    int L = len * ncomp;
    float *acc = (float *)calloc((size_t)ny * L, sizeof(float));
    for (int x = 0; x < nx; ++x)
        for (int y = 0; y < ny; ++y) {
            const float *v = arr + ((size_t)x * ny + y) * L;
            float *a = acc + (size_t)y * L;
            for (int i = 0; i < L; ++i) a[i] += v[i];
        }
    for (size_t i = 0; i < (size_t)ny * L; ++i) acc[i] = acc[i] / (float)nx;
    for (int i = 0; i < L; ++i) out[i] = 0.0f;
    for (int y = 0; y < ny; ++y)
        for (int i = 0; i < L; ++i) out[i] += acc[(size_t)y * L + i];
    for (int i = 0; i < L; ++i) out[i] = out[i] / (float)ny;
    free(acc);
}

/* ---------------------------------------------------------------------------
 * ROI — src/math_tools.rs:574-661.  usize arithmetic restated in uint64_t
 * with release-mode wrapping (a10); *would_panic is set when a debug build
 * would have panicked on underflow/overflow or divided by zero.
 * ------------------------------------------------------------------------- */
int thz_oracle_point_in_polygon(uint64_t x, uint64_t y, const uint64_t *poly_xy, int n,
                                int *would_panic)
{
    int inside = 0;
    int j = n - 1;
    for (int i = 0; i < n; ++i) {
        uint64_t xi = poly_xy[2 * i], yi = poly_xy[2 * i + 1];
        uint64_t xj = poly_xy[2 * j], yj = poly_xy[2 * j + 1];
        int intersect = 0;
        if ((yi > y) != (yj > y)) {
            /* short-circuit: the arithmetic only runs when the edge straddles y,
             * which also guarantees yj != yi (no division by zero) */
            if (would_panic && (xj < xi || y < yi || yj < yi)) *would_panic = 1;
            uint64_t num = (xj - xi) * (y - yi);
            uint64_t den = (yj - yi);
            uint64_t rhs = num / den + xi;
            intersect = x < rhs;
        }
        if (intersect) inside = !inside;
        j = i;
    }
    return inside;
}

/* mask over (y,x) in the function's own coordinate system: mask[y*x_size+x] */
void thz_oracle_roi_mask(const uint64_t *poly_xy_in, int n, uint64_t scaling, int shape0, int shape1,
                         uint8_t *mask, int *would_panic)
{
    uint64_t *poly = (uint64_t *)malloc(sizeof(uint64_t) * 2 * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) {
        poly[2 * i] = poly_xy_in[2 * i] / scaling;
        poly[2 * i + 1] = poly_xy_in[2 * i + 1] / scaling;
    }
    uint64_t x_size = (uint64_t)shape1, y_size = (uint64_t)shape0;
    memset(mask, 0, (size_t)shape0 * shape1);
    uint64_t x_min = UINT64_MAX, y_min = UINT64_MAX, x_max = 0, y_max = 0;
    for (int i = 0; i < n; ++i) {
        uint64_t x = poly[2 * i], y = poly[2 * i + 1];
        if (x < x_min) x_min = x;
        if (y < y_min) y_min = y;
        if (x > x_max) x_max = x;
        if (y > y_max) y_max = y;
    }
    if (x_min > x_size - 1) x_min = x_size - 1;
    if (y_min > y_size - 1) y_min = y_size - 1;
    if (x_max > x_size - 1) x_max = x_size - 1;
    if (y_max > y_size - 1) y_max = y_size - 1;
    if (n > 0)
        for (uint64_t y = y_min; y <= y_max; ++y)
            for (uint64_t x = x_min; x <= x_max; ++x)
                if (thz_oracle_point_in_polygon(x, y, poly, n, would_panic))
                    mask[y * x_size + x] = 1;
    free(poly);
}

/* math_tools.rs:599-661; data is (shape0, shape1, len) C-order */
void thz_oracle_average_polygon_roi(const float *data, int shape0, int shape1, int len,
                                    const uint64_t *poly_xy, int n, uint64_t scaling, float *out,
                                    int *would_panic)
{
    uint8_t *mask = (uint8_t *)malloc((size_t)shape0 * shape1 + 1);
    thz_oracle_roi_mask(poly_xy, n, scaling, shape0, shape1, mask, would_panic);
    for (int z = 0; z < len; ++z) out[z] = 0.0f;
    long count = 0;
    for (int y = 0; y < shape0; ++y)
        for (int x = 0; x < shape1; ++x)
            if (mask[(size_t)y * shape1 + x]) {
                const float *v = data + ((size_t)(shape0 - y - 1) * shape1 + x) * len;
                for (int z = 0; z < len; ++z) out[z] += v[z];
                ++count;
            }
    if (count > 0)
        for (int z = 0; z < len; ++z) out[z] /= (float)count;
    free(mask);
}

/* ---------------------------------------------------------------------------
 * Stage: ifft — src/math_tools.rs:418-571 (per-pixel part :545-568 and the
 * pixel means :421-440; ROI means are thz_oracle_average_polygon_roi).
 * Returns the number of pixels for which realfft would have returned Err
 * (non-zero imaginary part in bin 0 or, for even nt, the last bin; a'-4).
 * ------------------------------------------------------------------------- */
long thz_oracle_ifft_stage(const float *fft_interleaved, int nx, int ny, int nt, float *data_out,
                           int nthreads)
{
    int nf = nt / 2 + 1;
    rplan_f *r = rplan_new_f(nt);
    long nerr = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel reduction(+ : nerr)
#endif
    {
        cpx_f *work = (cpx_f *)malloc(sizeof(cpx_f) * (size_t)(2 * nt + 4));
        float *real = (float *)malloc(sizeof(float) * (size_t)nt);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int x = 0; x < nx; ++x) {
            for (int y = 0; y < ny; ++y) {
                size_t p = (size_t)x * ny + y;
                const cpx_f *spec = (const cpx_f *)(fft_interleaved + p * nf * 2);
                if (spec[0].im != 0.0f || (nt % 2 == 0 && spec[nf - 1].im != 0.0f)) ++nerr;
                irfft_f(r, spec, real, work);
                float *o = data_out + p * nt;
                for (int i = 0; i < nt; ++i) o[i] = real[i] / (float)nt; /* :563-565 */
            }
        }
        free(work); free(real);
    }
    rplan_free_f(r);
    return nerr;
}

/* avg_in_fourier_space path, math_tools.rs:442-470 / 496-529:
 * spectrum = from_polar(amp, phase) (bin 0 imaginary forced to 0 on the ROI
 * path only, :510-512), C2R, /nt. */
void thz_oracle_polar_irfft(const float *amp, const float *phase, int nt, int zero_dc_imag, float *out)
{
    int nf = nt / 2 + 1;
    cpx_f *spec = (cpx_f *)malloc(sizeof(cpx_f) * (size_t)nf);
    for (int k = 0; k < nf; ++k) {
        spec[k].re = amp[k] * cosf(phase[k]); /* Complex::from_polar */
        spec[k].im = amp[k] * sinf(phase[k]);
    }
    if (zero_dc_imag && nf > 0) spec[0].im = 0.0f;
    thz_oracle_irfft_f32((const float *)spec, nt, out);
    for (int i = 0; i < nt; ++i) out[i] = out[i] / (float)nt;
    free(spec);
}

/* ---------------------------------------------------------------------------
 * scaling — src/math_tools.rs:242-310 (scale_3d helper :273-301).
 * arr (nx,ny,len*ncomp) -> out (nx/s, ny/s, len*ncomp); divides by s*s even
 * on ragged edges (a'-6).
 * ------------------------------------------------------------------------- */
void thz_oracle_scale3d(const float *arr, int nx, int ny, int len, int ncomp, int s, float *out)
{
    int L = len * ncomp;
    int nw = nx / s, nh = ny / s;
    float sf = (float)(s * s);
    for (int ax = 0; ax < nw; ++ax)
        for (int ay = 0; ay < nh; ++ay)
            for (int z = 0; z < L; ++z) {
                float sum = 0.0f;
                for (int i = 0; i < s; ++i)
                    for (int j = 0; j < s; ++j) {
                        int ox = ax * s + i, oy = ay * s + j;
                        if (ox < nx && oy < ny) sum += arr[((size_t)ox * ny + oy) * L + z];
                    }
                out[((size_t)ax * nh + ay) * L + z] = sum / sf;
            }
}

/* ---------------------------------------------------------------------------
 * Tilt compensation — src/filters/tilt_compensation.rs:97-226 (K11).
 * Returns num_steps; when out != NULL fills new_time (nt + 2*num_steps) and
 * the extended cube out (nx, ny, nt + 2*num_steps).  dt is the reference's
 * hard-coded 0.05 (a'-7); ndarray::linspace = start + step*i.
 * ------------------------------------------------------------------------- */
static void oracle_linspace(float a, float b, int n, float *out)
{
    float step = n > 1 ? (b - a) / (float)(n - 1) : 0.0f;
    for (int i = 0; i < n; ++i) out[i] = a + step * (float)i;
}

int thz_oracle_tilt(const float *data, const float *time, int nx, int ny, int nt, double tilt_x_deg,
                    double tilt_y_deg, float dx, float dy, float *new_time, float *out)
{
    const float time_shift_x = (float)tilt_x_deg / 180.0f * PI_F; /* :105 */
    const float time_shift_y = (float)tilt_y_deg / 180.0f * PI_F;
    const int width = nx, height = ny;
    const float center_x = (float)width / 2.0f * dx;              /* :115 */
    const float center_y = (float)height / 2.0f * dy;
    const double c = 0.299792458;
    const float dt = 0.05f;
    const float max_offset_x = (float)((double)center_x * (double)fabsf(time_shift_x) / c);
    const float max_offset_y = (float)((double)center_y * (double)fabsf(time_shift_y) / c);
    float extension = (max_offset_x + max_offset_y) / dt;
    extension = floorf(extension) * dt;
    const int num_steps = (int)roundf(extension / dt);
    const int ext = nt + 2 * num_steps;
    if (!new_time || !out) return num_steps;
    const float first = time[0], last = time[nt - 1];
    oracle_linspace(first - extension, first - dt, num_steps, new_time);
    memcpy(new_time + num_steps, time, sizeof(float) * (size_t)nt);
    oracle_linspace(last + dt, last + extension, num_steps, new_time + num_steps + nt);
    float *win = (float *)malloc(sizeof(float) * (size_t)nt);
    for (int i = 0; i < width; ++i)
        for (int j = 0; j < height; ++j) {
            const float x_offset =
                (float)((double)(((float)i - (float)width / 2.0f) * dx) * (double)time_shift_x / c);
            const float y_offset =
                (float)((double)(((float)j - (float)height / 2.0f) * dy) * (double)time_shift_y / c);
            const float delta = x_offset + y_offset;
            const long delta_steps = (long)floorf(delta / dt);
            long ins = (long)num_steps + delta_steps;
            if (ins < 0) ins = 0;
            const float *raw = data + ((size_t)i * height + j) * nt;
            float *e = out + ((size_t)i * height + j) * ext;
            long end = ins + nt;
            if (end > ext) end = ext;
            if (ins > ext) ins = ext; /* Rust would panic on the slice; not reachable for |tilt| <= 15 deg */
            for (long k = 0; k < ins; ++k) e[k] = raw[0];
            memcpy(win, raw, sizeof(float) * (size_t)nt);
            thz_oracle_apply_adapted_blackman(win, time, nt, 0.0f, 7.0f);
            for (long k = ins; k < end; ++k) e[k] = win[k - ins];
            for (long k = end; k < ext; ++k) e[k] = 0.0f;
        }
    free(win);
    return num_steps;
}

/* ---------------------------------------------------------------------------
 * CPU baseline: the default chain, stage-fused per trace without per-stage
 * container copies (SURVEY §8d variant (i)), OpenMP over Axis(0) like the
 * reference's rayon split.  Composite of the functions above.
 *   in   : data (nx,ny,nt) raw (already bias-subtracted)
 *   out  : fft (band-passed), amplitudes (band-passed), phases, data_out, img
 * ------------------------------------------------------------------------- */
void thz_oracle_pipeline(const float *data_in, const float *time, int nx, int ny, int nt,
                         const float *w_tilt /* may be NULL */, const float *w_td_before,
                         int window_type, float win_lo, float win_hi,
                         const float *w_fd, const float *w_td_after, float *fft_interleaved,
                         float *amplitudes, float *phases, float *data_out, float *img, int nthreads)
{
    int nf = nt / 2 + 1;
    rplan_f *r = rplan_new_f(nt);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
#endif
    {
        cpx_f *work = (cpx_f *)malloc(sizeof(cpx_f) * (size_t)(2 * nt + 4));
        cpx_f *spec = (cpx_f *)malloc(sizeof(cpx_f) * (size_t)nf);
        float *ph = (float *)malloc(sizeof(float) * (size_t)nf);
        float *tr = (float *)malloc(sizeof(float) * (size_t)nt);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int x = 0; x < nx; ++x) {
            for (int y = 0; y < ny; ++y) {
                size_t p = (size_t)x * ny + y;
                const float *src = data_in + p * nt;
                /* tilt taper (tilt_compensation.rs:186-188), then Time Band Pass
                 * (band_pass_td_before_fft.rs:155-175), then the fft window */
                if (w_tilt) for (int i = 0; i < nt; ++i) tr[i] = src[i] * w_tilt[i];
                else for (int i = 0; i < nt; ++i) tr[i] = src[i];
                for (int i = 0; i < nt; ++i) tr[i] = tr[i] * w_td_before[i];
                thz_oracle_apply_window(window_type, tr, time, nt, win_lo, win_hi);
                rfft_f(r, tr, spec, work);
                float *amp = amplitudes + p * nf;
                for (int k = 0; k < nf; ++k) {
                    amp[k] = hypotf(spec[k].re, spec[k].im);
                    ph[k] = atan2f(spec[k].im, spec[k].re);
                }
                thz_oracle_numpy_unwrap(ph, nf, 2.0f * PI_F, phases + p * nf);
                for (int k = 0; k < nf; ++k) {
                    spec[k].re = spec[k].re * w_fd[k];
                    spec[k].im = spec[k].im * w_fd[k];
                    amp[k] = amp[k] * w_fd[k];
                }
                memcpy(fft_interleaved + p * nf * 2, spec, sizeof(cpx_f) * (size_t)nf);
                irfft_f(r, spec, tr, work);
                float *o = data_out + p * nt;
                float s = 0.0f;
                for (int i = 0; i < nt; ++i) {
                    float v = (tr[i] / (float)nt) * w_td_after[i];
                    o[i] = v;
                    s += v * v;
                }
                img[p] = s;
            }
        }
        free(work); free(spec); free(ph); free(tr);
    }
    rplan_free_f(r);
}

/* ConfigCommand::OpenRef, data_thread.rs:372-588: index-shift alignment of a reference pulse
 * to the scan's time axis (:405-481), window with the reference file's own time axis
 * (:490-515; zip stops at the shorter of the two), transform with the scan's plan, |X| and
 * unwrapped phase (:517-533).  Returns 0; -1 where the reference would panic (ndarray Zip of
 * unequal lengths in the non-adapted windows); align_mode: 0 untouched, 1 shifted, 2 naive. */
int thz_oracle_open_ref(const float *scan_time, int nt, const float *ref_time, const float *ref_signal, int nref,
                        int window_type, float win_lo, float win_hi, float *reference_out, float *amp_out,
                        float *phase_out, int *align_mode)
{
    float *ref = reference_out;
    int mode = 0;
    if (nt != nref || (nref > 0 && fabsf(scan_time[0] - ref_time[0]) > 1e-9f)) {
        if (nt > 1 && nref > 1) {
            mode = 1;
            for (int i = 0; i < nt; ++i) ref[i] = 0.0f;
            const float ref_dt = ref_time[1] - ref_time[0];
            const float time_offset = scan_time[0] - ref_time[0];
            const float q = roundf(time_offset / ref_dt);
            long index_offset;
            if (isnan(q)) index_offset = 0;            /* Rust `as isize`: NaN -> 0, saturating */
            else if (q >= 9.2e18f) index_offset = LONG_MAX;
            else if (q <= -9.2e18f) index_offset = LONG_MIN;
            else index_offset = (long)q;
            size_t src_start = index_offset > 0 ? (size_t)index_offset : 0;
            size_t dst_start = index_offset < 0 ? (size_t)(-(index_offset + 1)) + 1 : 0;
            size_t src_len = (size_t)nref > src_start ? (size_t)nref - src_start : 0;
            size_t dst_len = (size_t)nt > dst_start ? (size_t)nt - dst_start : 0;
            size_t copy_len = src_len < dst_len ? src_len : dst_len;
            for (size_t i = 0; i < copy_len; ++i) ref[dst_start + i] = ref_signal[src_start + i];
        } else {
            mode = 2;
            for (int i = 0; i < nt; ++i) ref[i] = i < nref ? ref_signal[i] : 0.0f;
        }
    } else {
        for (int i = 0; i < nt; ++i) ref[i] = ref_signal[i];
    }
    if (align_mode) *align_mode = mode;
    if (window_type == WIN_ADAPTED_BLACKMAN) {
        const int m = nt < nref ? nt : nref;
        if (nref > 0) {
            const float t0 = ref_time[0], tn = ref_time[nref - 1];
            for (int i = 0; i < m; ++i) {
                const float t = ref_time[i];
                if (t <= win_lo + t0) ref[i] *= thz_oracle_blackman_window(t - t0, 2.0f * win_lo);
                else if (t >= tn - win_hi) ref[i] *= thz_oracle_blackman_window(t - (tn - win_hi * 2.0f), 2.0f * win_hi);
            }
        }
    } else {
        if (nt != nref) return -1;
        thz_oracle_apply_window(window_type, ref, ref_time, nt, win_lo, win_hi);
    }
    const int nf = nt / 2 + 1;
    rplan_f *r = rplan_new_f(nt);
    cpx_f *spec = (cpx_f *)malloc(sizeof(cpx_f) * (size_t)nf);
    cpx_f *work = (cpx_f *)malloc(sizeof(cpx_f) * (size_t)(2 * nt + 4));
    float *ph = (float *)malloc(sizeof(float) * (size_t)nf);
    rfft_f(r, ref, spec, work);
    for (int k = 0; k < nf; ++k) {
        amp_out[k] = hypotf(spec[k].re, spec[k].im);
        ph[k] = atan2f(spec[k].im, spec[k].re);
    }
    thz_oracle_numpy_unwrap(ph, nf, 2.0f * PI_F, phase_out);
    free(ph); free(work); free(spec);
    rplan_free_f(r);
    return 0;
}

/* calculate_optical_properties, math_tools.rs:663-701: refractive index, absorption
 * and extinction coefficient per frequency bin from sample / reference spectra */
void thz_oracle_optical_properties(const float *sample_amp, const float *sample_phase, const float *ref_amp,
                                   const float *ref_phase, const float *freq, size_t nf, float thickness,
                                   float *n_out, float *alpha_out, float *kappa_out)
{
    const float C = 2.99792458e8f;
    for (size_t i = 0; i < nf; ++i) {
        const float frequency_hz = freq[i] * 1.0e12f;
        const float delta_phi = sample_phase[i] - ref_phase[i];
        const float omega = 2.0f * PI_F * frequency_hz;
        const float n = 1.0f + C * delta_phi / (omega * thickness);
        const float amp = fmaxf(sample_amp[i], 1e-12f);
        const float amp_ref = fmaxf(ref_amp[i], 1e-12f);
        const float n_safe = fmaxf(n, 1e-6f);
        const float np1 = n_safe + 1.0f;
        const float alpha = -2.0f / thickness * logf((np1 * np1) / (4.0f * n_safe) * amp / amp_ref);
        const float kappa = alpha * C / (4.0f * PI_F * frequency_hz);
        n_out[i] = n;
        alpha_out[i] = alpha;
        kappa_out[i] = kappa;
    }
}

#include "thz_oracle_deconv.c"
#include "thz_oracle_voxel.c"

int thz_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
