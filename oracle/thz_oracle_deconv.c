/* thz_oracle_deconv.c — CPU restatement of the reference's frequency-dependent
 * Richardson–Lucy deconvolution (K12), src/filters/deconvolution.rs and
 * src/filters/psf.rs.  TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Included by thz_oracle.c (shares its FFT instantiations).
 *
 * PARITY: the reference never executes this math in its own tests (its only
 * deconvolution test returns early at MIN_IMAGE_SIZE, deconvolution.rs:802-812),
 * so everything here is pinned only by being a line-by-line restatement.
 * Stated divergences:
 *  - convolve2d's FFT path (deconvolution.rs:489-545) is restated as the
 *    mathematically intended "same" linear convolution, evaluated directly;
 *    the reference's plan-size/buffer-size mismatch is not
 *    reproduced.  The <= 256-element path is the reference's
 *    correlation-indexed sum, verbatim (:432-458).
 *  - interp1d 0.2.0 (un-vendored) is restated as plain linear interpolation
 *    y0 + (y1-y0)*(x-x0)/(x1-x0) on the sorted knots.
 *  - the band sum is accumulated in band order (rayon's reduce order is
 *    nondeterministic in the reference, :1008-1013).
 */

typedef struct thz_spline_c {
    int n;                /* number of knots */
    const float *knots;   /* n   */
    const float *values;  /* n   */
    const float *a, *b, *c, *d; /* n-1 each */
} thz_spline_c;

typedef struct thz_hybrid_c {
    float base_a, base_b;
    thz_spline_c corr;
} thz_hybrid_c;

typedef struct thz_psf_c {
    thz_hybrid_c wx, wy;
    thz_spline_c x0, y0;
} thz_psf_c;

/* psf.rs:26-80 */
static float spline_eval_single(const thz_spline_c *s, float x)
{
    int n = s->n;
    if (n == 0) return 0.0f;
    if (x < s->knots[0]) {
        float dx = x - s->knots[0];
        float y = s->a[0] + s->b[0] * dx;
        return fmaxf(y, 1e-6f);
    }
    if (x > s->knots[n - 1]) {
        int i = n - 2;
        float dxe = s->knots[n - 1] - s->knots[i];
        float y_end = s->a[i] + s->b[i] * dxe + s->c[i] * dxe * dxe + s->d[i] * dxe * dxe * dxe;
        float slope = s->b[i] + 2.0f * s->c[i] * dxe + 3.0f * s->d[i] * dxe * dxe;
        float dx = x - s->knots[n - 1];
        return fmaxf(y_end + slope * dx, 1e-6f);
    }
    int left = 0, right = n - 1;
    while (right - left > 1) {
        int mid = (left + right) / 2;
        if (s->knots[mid] > x) right = mid; else left = mid;
    }
    float dx = x - s->knots[left];
    return s->a[left] + s->b[left] * dx + s->c[left] * dx * dx + s->d[left] * dx * dx * dx;
}

/* psf.rs:83-117 */
static float spline_eval_const_extrap(const thz_spline_c *s, float x)
{
    int n = s->n;
    if (n == 0) return 0.0f;
    if (x < s->knots[0]) return s->values[0];
    if (x > s->knots[n - 1]) return s->values[n - 1];
    int left = 0, right = n - 1;
    while (right - left > 1) {
        int mid = (left + right) / 2;
        if (s->knots[mid] > x) right = mid; else left = mid;
    }
    float dx = x - s->knots[left];
    return s->a[left] + s->b[left] * dx + s->c[left] * dx * dx + s->d[left] * dx * dx * dx;
}

/* psf.rs:134-179 */
static float hybrid_eval_correction(const thz_hybrid_c *h, float f)
{
    const thz_spline_c *s = &h->corr;
    int n = s->n;
    if (n == 0) return 0.0f;
    float f_min = s->knots[0], f_max = s->knots[n - 1];
    if (f >= f_min && f <= f_max) return spline_eval_single(s, f);
    if (f < f_min) {
        float dx = f - f_min;
        float y0 = s->a[0], slope = s->b[0];
        float max_slope = h->base_a / (f * f);
        float safe = fminf(slope, max_slope);
        return y0 + safe * dx;
    } else {
        int i = n - 2;
        float dxe = s->knots[n - 1] - s->knots[i];
        float y_end = s->a[i] + s->b[i] * dxe + s->c[i] * dxe * dxe + s->d[i] * dxe * dxe * dxe;
        float slope_end = s->b[i] + 2.0f * s->c[i] * dxe + 3.0f * s->d[i] * dxe * dxe;
        float max_slope = h->base_a / (f * f);
        float safe = fminf(slope_end, max_slope);
        float dx = f - s->knots[n - 1];
        return y_end + safe * dx;
    }
}

/* psf.rs:122-131 */
static float hybrid_eval_single(const thz_hybrid_c *h, float f)
{
    float base = h->base_a / f + h->base_b;
    float corr = hybrid_eval_correction(h, f);
    return fmaxf(base + corr, 1e-6f);
}

void thz_oracle_psf_eval(const thz_psf_c *psf, const float *freqs, int n, float *wx, float *wy,
                         float *x0, float *y0)
{
    for (int i = 0; i < n; ++i) {
        wx[i] = hybrid_eval_single(&psf->wx, freqs[i]);
        wy[i] = hybrid_eval_single(&psf->wy, freqs[i]);
        x0[i] = spline_eval_const_extrap(&psf->x0, freqs[i]);
        y0[i] = spline_eval_const_extrap(&psf->y0, freqs[i]);
    }
}

/* ---- FIR bank, deconvolution.rs:30-211 (f64) ---------------------------- */
static double kaiser_atten(int ntaps, double width_ratio)
{
    double a = 2.285 * ((double)ntaps - 1.0) * M_PI * width_ratio + 7.95;
    return a > 0.0 ? a : 0.0;
}
static double kaiser_beta(double atten)
{
    if (atten > 50.0) return 0.1102 * (atten - 8.7);
    if (atten >= 21.0) return 0.5842 * pow(atten - 21.0, 0.4) + 0.07886 * (atten - 21.0);
    return 0.0;
}
static double bessel_i0(double x)
{
    double sum = 1.0, term = 1.0;
    double xh = (x / 2.0) * (x / 2.0);
    for (int k = 1; k < 50; ++k) {
        term *= xh / (double)(k * k);
        sum += term;
        if (term < 1e-12 * sum) break;
    }
    return sum;
}
static double sinc_d(double x) { return fabs(x) < 1e-10 ? 1.0 : sin(x) / x; }
static double kaiser_coeff(int n, int n_taps, double beta)
{
    if (n == 0 || n == n_taps - 1) return 0.0;
    double arg = 2.0 * (double)n / ((double)n_taps - 1.0) - 1.0;
    return bessel_i0(beta * sqrt(1.0 - arg * arg)) / bessel_i0(beta);
}
/* out has n_taps entries */
static void firwin_lowpass(int n_taps, double cutoff_hz, double beta, double fs, double *out)
{
    int adj = (n_taps % 2 == 0) ? n_taps - 1 : n_taps;
    double mid = (double)(adj - 1) / 2.0;
    double cutoff = cutoff_hz / fs;
    double sum = 0.0;
    for (int n = 0; n < adj; ++n) {
        out[n] = sinc_d(2.0 * M_PI * cutoff * ((double)n - mid)) * kaiser_coeff(n, adj, beta);
        sum += out[n];
    }
    if (fabs(sum) > 1e-10)
        for (int n = 0; n < adj; ++n) out[n] /= sum;
    if (n_taps % 2 == 0) out[adj] = 0.0;
}
static void firwin_highpass(int n_taps, double cutoff_hz, double beta, double fs, double *out)
{
    int adj = (n_taps % 2 == 0) ? n_taps - 1 : n_taps;
    double mid = (double)(adj - 1) / 2.0;
    firwin_lowpass(adj, cutoff_hz, beta, fs, out);
    for (int i = 0; i < adj; ++i) out[i] = (i == (int)mid) ? 1.0 - out[i] : -out[i];
    if (n_taps % 2 == 0) out[adj] = 0.0;
}
static void bandpass_kaiser(int ntaps, double lowcut, double highcut, double fs, double width, double *out)
{
    double beta = kaiser_beta(kaiser_atten(ntaps, width / (0.5 * fs)));
    if (lowcut <= 0.0) {
        firwin_lowpass(ntaps, highcut, beta, fs, out);
    } else if (highcut >= 0.5 * fs) {
        firwin_highpass(ntaps, lowcut, beta, fs, out);
    } else {
        double *hl = (double *)malloc(sizeof(double) * 2 * (size_t)ntaps);
        double *hh = hl + ntaps;
        firwin_highpass(ntaps, lowcut, beta, fs, hl);
        firwin_highpass(ntaps, highcut, beta, fs, hh);
        for (int i = 0; i < ntaps; ++i) out[i] = hl[i] - hh[i];
        free(hl);
    }
}

#define THZ_DECONV_NTAPS 499

/* deconvolution.rs:160-211: filters (n_filters x 499) f32, centers f32 */
void thz_oracle_filter_bank(int n_filters, double start_freq, double end_freq, double win_width,
                            const float *time, float *filters, float *centers)
{
    const int ntaps = THZ_DECONV_NTAPS;
    double dt = (double)(time[1] - time[0]);
    double fs = 1.0 / dt;
    double log_start = log(start_freq), log_end = log(end_freq);
    double log_step = (log_end - log_start) / (double)(n_filters - 1);
    for (int i = 0; i < n_filters; ++i) centers[i] = (float)exp(log_start + (double)i * log_step);
    double *h = (double *)malloc(sizeof(double) * (size_t)ntaps);
    for (int i = 0; i < n_filters; ++i) {
        double cf = (double)centers[i];
        double lowcut = (i == 0) ? 0.0 : sqrt((double)centers[i - 1] * cf);
        double highcut = (i == n_filters - 1) ? 0.5 * fs : sqrt(cf * (double)centers[i + 1]);
        bandpass_kaiser(ntaps, lowcut, highcut, fs, win_width, h);
        for (int j = 0; j < ntaps; ++j) filters[(size_t)i * ntaps + j] = (float)h[j];
    }
    free(h);
}

/* ---- PSF construction, psf.rs:228-332 ----------------------------------- */
/* psf.rs:326-332 */
static float gaussian_f(float xi, float x0, float w)
{
    float d = xi - x0;
    return sqrtf(2.0f / PI_F) * expf(-2.0f * (d * d) / (w * w)) / w;
}

/* linear interpolation on sorted knots (interp1d 0.2.0 stand-in) */
static float interp_lin(const float *xs, const float *ys, int n, float x)
{
    if (n == 1) return ys[0];
    int lo = 0, hi = n - 1;
    if (x <= xs[0]) { lo = 0; hi = 1; }
    else if (x >= xs[n - 1]) { lo = n - 2; hi = n - 1; }
    else {
        while (hi - lo > 1) {
            int mid = (lo + hi) / 2;
            if (xs[mid] > x) hi = mid; else lo = mid;
        }
    }
    float x0 = xs[lo], x1 = xs[hi];
    return ys[lo] + (ys[hi] - ys[lo]) * (x - x0) / (x1 - x0);
}

/* psf.rs:228-313.  Inputs: 1-D profiles on coordinates x (len nxp) and y (len nyp).
 * Output dims (*rows = 2*x_max+1, *cols = 2*y_max+1); psf may be NULL for sizing. */
void thz_oracle_create_psf_2d(const float *psf_x_in, const float *psf_y_in, const float *x_in,
                              const float *y_in, int nxp, int nyp, float dx, float dy, float *psf,
                              int *rows, int *cols)
{
    float mx = -FLT_MAX, my = -FLT_MAX, cxm = -FLT_MAX, cym = -FLT_MAX;
    for (int i = 0; i < nxp; ++i) { mx = fmaxf(mx, psf_x_in[i]); cxm = fmaxf(cxm, x_in[i]); }
    for (int i = 0; i < nyp; ++i) { my = fmaxf(my, psf_y_in[i]); cym = fmaxf(cym, y_in[i]); }
    long x_max = (long)floorf(cxm), y_max = (long)floorf(cym);
    if (x_max < 0) x_max = 0; /* `as usize` saturates */
    if (y_max < 0) y_max = 0;
    *rows = (int)(2 * x_max + 1);
    *cols = (int)(2 * y_max + 1);
    if (!psf) return;
    float new_x_max = ceilf(2.0f * (float)x_max), new_y_max = ceilf(2.0f * (float)y_max);
    float x_step = x_in[nxp - 1] - x_in[nxp - 2], y_step = y_in[nyp - 1] - y_in[nyp - 2];
    float fx = ceilf((new_x_max - x_in[nxp - 1]) / x_step), fy = ceilf((new_y_max - y_in[nyp - 1]) / y_step);
    int ex = fx > 0 ? (int)fx : 0, ey = fy > 0 ? (int)fy : 0; /* `as usize` saturates at 0 */
    int nxt = nxp + 2 * ex, nyt = nyp + 2 * ey;
    float *xs = (float *)malloc(sizeof(float) * (size_t)(2 * nxt + 2 * nyt));
    float *px = xs + nxt, *ys = px + nxt, *py = ys + nyt;
    for (int i = 0; i < nxp; ++i) { xs[ex + i] = x_in[i]; px[ex + i] = psf_x_in[i] / mx; }
    for (int i = 0; i < nyp; ++i) { ys[ey + i] = y_in[i]; py[ey + i] = psf_y_in[i] / my; }
    /* each loop round appends x[last] + step and prepends x[0] - step (f32, sequential) */
    for (int k = 0; k < ex; ++k) {
        xs[ex + nxp + k] = xs[ex + nxp + k - 1] + x_step; px[ex + nxp + k] = 0.0f;
        xs[ex - 1 - k] = xs[ex - k] - x_step;             px[ex - 1 - k] = 0.0f;
    }
    for (int k = 0; k < ey; ++k) {
        ys[ey + nyp + k] = ys[ey + nyp + k - 1] + y_step; py[ey + nyp + k] = 0.0f;
        ys[ey - 1 - k] = ys[ey - k] - y_step;             py[ey - 1 - k] = 0.0f;
    }
    for (long i = -x_max; i <= x_max; ++i)
        for (long j = -y_max; j <= y_max; ++j) {
            float a = interp_lin(xs, px, nxt, (float)i * dx);
            float b = interp_lin(ys, py, nyt, (float)j * dy);
            psf[(size_t)(i + x_max) * (size_t)(*cols) + (size_t)(j + y_max)] = a * b;
        }
    free(xs);
}

/* Per-band PSF as Deconvolution::filter builds it, deconvolution.rs:906-960.
 * psf may be NULL (sizing). */
void thz_oracle_band_psf(const thz_psf_c *P, float center_freq, float dx, float dy, int img_rows,
                         int img_cols, float *psf, int *rows, int *cols, float *wx_out)
{
    float wx = hybrid_eval_single(&P->wx, center_freq);
    float wy = hybrid_eval_single(&P->wy, center_freq);
    float x0 = spline_eval_const_extrap(&P->x0, center_freq);
    float y0 = spline_eval_const_extrap(&P->y0, center_freq);
    if (wx_out) *wx_out = wx;
    float rx = (wx + fabsf(x0)) * 3.0f, ry = (wy + fabsf(y0)) * 3.0f;
    if (rx < 2.5f) rx = 2.5f;
    if (ry < 2.5f) ry = 2.5f;
    rx = floorf(rx / dx) * dx + dx;
    ry = floorf(ry / dy) * dy + dy;
    float max_x = ((float)img_cols - 2.0f) * dx / 2.0f, max_y = ((float)img_rows - 2.0f) * dy / 2.0f;
    float cx = fminf(rx, max_x), cy = fminf(ry, max_y);
    long kx = (long)floorf(cx / dx), ky = (long)floorf(cy / dy);
    int nxp = (int)(2 * kx + 1), nyp = (int)(2 * ky + 1);
    float *buf = (float *)malloc(sizeof(float) * (size_t)(2 * nxp + 2 * nyp));
    float *xv = buf, *gx = buf + nxp, *yv = gx + nxp, *gy = yv + nyp;
    for (long i = -kx; i <= kx; ++i) { xv[i + kx] = (float)i * dx; gx[i + kx] = gaussian_f(xv[i + kx], x0, wx); }
    for (long i = -ky; i <= ky; ++i) { yv[i + ky] = (float)i * dy; gy[i + ky] = gaussian_f(yv[i + ky], y0, wy); }
    thz_oracle_create_psf_2d(gx, gy, xv, yv, nxp, nyp, dx, dy, psf, rows, cols);
    free(buf);
}

/* ---- convolutions ------------------------------------------------------- */
/* deconvolution.rs:266-317 (Complex<f64> FFT), "same" slice [shift, shift+na) */
static void convolve1d_f64(const float *a, int na, const float *b, int nb, const plan_d *pl,
                           int fft_size, cpx_d *wa, cpx_d *wb, cpx_d *tmp, float *out)
{
    for (int i = 0; i < fft_size; ++i) { wa[i].re = wa[i].im = 0.0; wb[i].re = wb[i].im = 0.0; }
    for (int i = 0; i < na; ++i) wa[i].re = (double)a[i];
    for (int i = 0; i < nb; ++i) wb[i].re = (double)b[i];
    cfft_fwd_d(pl, wa, tmp); memcpy(wa, tmp, sizeof(cpx_d) * (size_t)fft_size);
    cfft_fwd_d(pl, wb, tmp); memcpy(wb, tmp, sizeof(cpx_d) * (size_t)fft_size);
    for (int i = 0; i < fft_size; ++i) wa[i] = cmul_d(wa[i], wb[i]);
    cfft_inv_d(pl, wa, tmp);
    int shift = (nb - 1) / 2;
    for (int i = 0; i < na; ++i) out[i] = (float)(tmp[i + shift].re / (double)fft_size);
}

/* DIAGNOSTIC ONLY (scripts/gpu_deconv_error_budget.py): the same convolution through an f32 FFT — NOT the
 * reference's arithmetic (deconvolution.rs:266-317 is Complex<f64>), used to measure how much of the device's
 * distance from the oracle is the fp32 FIR and how much the Richardson-Lucy iterations add on top. */
static int g_fir_f32 = 0;
void thz_oracle_set_fir_f32(int on) { g_fir_f32 = on; }
static void convolve1d_f32(const float *a, int na, const float *b, int nb, const plan_f *pl,
                           int fft_size, cpx_f *wa, cpx_f *wb, cpx_f *tmp, float *out)
{
    for (int i = 0; i < fft_size; ++i) { wa[i].re = wa[i].im = 0.0f; wb[i].re = wb[i].im = 0.0f; }
    for (int i = 0; i < na; ++i) wa[i].re = a[i];
    for (int i = 0; i < nb; ++i) wb[i].re = b[i];
    cfft_fwd_f(pl, wa, tmp); memcpy(wa, tmp, sizeof(cpx_f) * (size_t)fft_size);
    cfft_fwd_f(pl, wb, tmp); memcpy(wb, tmp, sizeof(cpx_f) * (size_t)fft_size);
    for (int i = 0; i < fft_size; ++i) wa[i] = cmul_f(wa[i], wb[i]);
    cfft_inv_f(pl, wa, tmp);
    int shift = (nb - 1) / 2;
    for (int i = 0; i < na; ++i) out[i] = tmp[i + shift].re / (float)fft_size;
}

/* filter_scan, deconvolution.rs:574-609 */
void thz_oracle_filter_scan(const float *data, size_t npix, int nt, const float *filter, int ntaps,
                            float *out)
{
    int conv = nt + ntaps - 1, fft_size = 1;
    while (fft_size < conv) fft_size <<= 1;
    if (g_fir_f32) {
        plan_f *plf = plan_new_f(fft_size);
#ifdef _OPENMP
#pragma omp parallel
#endif
        {
            cpx_f *wa = (cpx_f *)malloc(sizeof(cpx_f) * 3 * (size_t)fft_size);
            cpx_f *wb = wa + fft_size, *tmp = wb + fft_size;
#ifdef _OPENMP
#pragma omp for
#endif
            for (long p = 0; p < (long)npix; ++p)
                convolve1d_f32(data + (size_t)p * nt, nt, filter, ntaps, plf, fft_size, wa, wb, tmp, out + (size_t)p * nt);
            free(wa);
        }
        plan_free_f(plf);
        return;
    }
    plan_d *pl = plan_new_d(fft_size);
#ifdef _OPENMP
#pragma omp parallel
#endif
    {
        cpx_d *wa = (cpx_d *)malloc(sizeof(cpx_d) * 3 * (size_t)fft_size);
        cpx_d *wb = wa + fft_size, *tmp = wb + fft_size;
#ifdef _OPENMP
#pragma omp for
#endif
        for (long p = 0; p < (long)npix; ++p)
            convolve1d_f64(data + (size_t)p * nt, nt, filter, ntaps, pl, fft_size, wa, wb, tmp,
                           out + (size_t)p * nt);
        free(wa);
    }
    plan_free_d(pl);
}

/* deconvolution.rs:432-458 verbatim (correlation-indexed, zero outside) */
/* Rows of the output are independent and every pixel keeps the reference's own summation order (m outer,
 * n inner, one f32 chain), so the OpenMP split over rows changes no bit of the result — it only lets the
 * reference's defaults (500 iterations, 47 x 57 taps) finish in seconds on the test box.  The team size is set
 * by the caller (thz_oracle_set_conv_threads): these loops run a thousand times per call, and a team of every
 * hardware thread of a 256-thread host inside a 16-CPU container spends its time in barriers. */
static int g_conv_threads = 8;
void thz_oracle_set_conv_threads(int n) { g_conv_threads = n > 0 ? n : 1; }
static void direct_convolve2d(const float *a, int ar, int ac, const float *b, int br, int bc, float *res)
{
    int hr = br / 2, hc = bc / 2;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(g_conv_threads)
#endif
    for (int i = 0; i < ar; ++i)
        for (int j = 0; j < ac; ++j) {
            float sum = 0.0f;
            for (int m = 0; m < br; ++m)
                for (int n = 0; n < bc; ++n) {
                    int x = i + m - hr, y = j + n - hc;
                    if (x >= 0 && y >= 0 && x < ar && y < ac) sum += a[(size_t)x * ac + y] * b[(size_t)m * bc + n];
                }
            res[(size_t)i * ac + j] = sum;
        }
}

/* the intended result of the FFT path (:489-545): full linear convolution,
 * rows/cols [(b-1)/2, (b-1)/2 + a) */
static void same_convolve2d(const float *a, int ar, int ac, const float *b, int br, int bc, float *res)
{
    int sr = (br - 1) / 2, sc = (bc - 1) / 2;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(g_conv_threads)
#endif
    for (int i = 0; i < ar; ++i)
        for (int j = 0; j < ac; ++j) {
            float sum = 0.0f;
            for (int m = 0; m < br; ++m)
                for (int n = 0; n < bc; ++n) {
                    int x = i + sr - m, y = j + sc - n;
                    if (x >= 0 && y >= 0 && x < ar && y < ac) sum += a[(size_t)x * ac + y] * b[(size_t)m * bc + n];
                }
            res[(size_t)i * ac + j] = sum;
        }
}

static void convolve2d_ref(const float *a, int ar, int ac, const float *b, int br, int bc, float *res)
{
    if (br * bc <= 256) direct_convolve2d(a, ar, ac, b, br, bc, res); /* THRESHOLD, :484-487 */
    else same_convolve2d(a, ar, ac, b, br, bc, res);
}

/* richardson_lucy, deconvolution.rs:620-712.  image (h, w) -> out (h, w) */
void thz_oracle_richardson_lucy(const float *image, int h, int w, const float *psf, int pr, int pc,
                                int n_iterations, float *out)
{
    int pad_y = pr / 2, pad_x = pc / 2;
    int H = h + 2 * pad_y, W = w + 2 * pad_x;
    size_t sz = (size_t)H * W;
    float *d = (float *)calloc(sz * 4, sizeof(float));
    float *u = d + sz, *t1 = u + sz, *t2 = t1 + sz;
    float *mirror = (float *)malloc(sizeof(float) * (size_t)pr * pc);
    for (int m = 0; m < pr; ++m)
        for (int n = 0; n < pc; ++n) mirror[(size_t)m * pc + n] = psf[(size_t)(pr - 1 - m) * pc + (pc - 1 - n)];
    for (int y = 0; y < h; ++y) memcpy(d + (size_t)(pad_y + y) * W + pad_x, image + (size_t)y * w, sizeof(float) * (size_t)w);
    for (int i = 0; i < pad_y; ++i) {
        memcpy(d + (size_t)i * W + pad_x, image + (size_t)(pad_y - i) * w, sizeof(float) * (size_t)w);
        memcpy(d + (size_t)(pad_y + h + i) * W + pad_x, image + (size_t)(h - 2 - i) * w, sizeof(float) * (size_t)w);
    }
    for (int j = 0; j < pad_x; ++j)
        for (int y = 0; y < H; ++y) {
            d[(size_t)y * W + j] = d[(size_t)y * W + pad_x + (pad_x - j)];
            d[(size_t)y * W + pad_x + w + j] = d[(size_t)y * W + pad_x + w - 2 - j];
        }
    memcpy(u, d, sizeof(float) * sz);
    const float eps = 1e-12f;
    for (int it = 0; it < n_iterations; ++it) {
        convolve2d_ref(u, H, W, psf, pr, pc, t1);
        for (size_t k = 0; k < sz; ++k) t1[k] = d[k] / (t1[k] + eps);
        convolve2d_ref(t1, H, W, mirror, pr, pc, t2);
        for (size_t k = 0; k < sz; ++k) u[k] *= t2[k];
    }
    for (int y = 0; y < h; ++y) memcpy(out + (size_t)y * w, u + (size_t)(pad_y + y) * W + pad_x, sizeof(float) * (size_t)w);
    free(d);
    free(mirror);
}

/* Deconvolution::filter, deconvolution.rs:766-1041.
 * Returns 0 on success, 1 when a guard makes the reference return the input
 * unchanged (out = in then).  gains_out (n_filters, nx, ny) optional. */
int thz_oracle_deconvolution(const float *data, const float *time, int nx, int ny, int nt, float dx,
                             float dy, const thz_psf_c *P, int n_iterations, int n_filters,
                             double start_freq, double end_freq, double win_width, float *out,
                             float *img_out, float *gains_out, int *n_iter_out)
{
    const int ntaps = THZ_DECONV_NTAPS;
    size_t npix = (size_t)nx * ny, cube = npix * (size_t)nt;
    memcpy(out, data, sizeof(float) * cube);
    if (P->wx.corr.n == 0) return 1;       /* :790 */
    if (nx < 16 || ny < 16) return 1;      /* :802-812 */
    float *filters = (float *)malloc(sizeof(float) * (size_t)n_filters * ntaps);
    float *centers = (float *)malloc(sizeof(float) * (size_t)n_filters * 3);
    float *wxv = centers + n_filters, *wyv = wxv + n_filters;
    thz_oracle_filter_bank(n_filters, start_freq, end_freq, win_width, time, filters, centers);
    float wx_min = INFINITY, wx_max = -INFINITY, wy_min = INFINITY, wy_max = -INFINITY;
    for (int i = 0; i < n_filters; ++i) {
        wxv[i] = hybrid_eval_single(&P->wx, centers[i]);
        wyv[i] = hybrid_eval_single(&P->wy, centers[i]);
        wx_min = fminf(wx_min, wxv[i]); wx_max = fmaxf(wx_max, wxv[i]);
        wy_min = fminf(wy_min, wyv[i]); wy_max = fmaxf(wy_max, wyv[i]);
    }
    float w_min = fminf(wx_min, wy_min), w_max = fmaxf(wx_max, wy_max);
    int img_rows = nx, img_cols = ny;
    long mpx = (long)ceilf(wx_max / dx) * 2 + 1, mpy = (long)ceilf(wy_max / dy) * 2 + 1;
    if (mpx < 3) mpx = 3;
    if (mpy < 3) mpy = 3;
    if (mpx >= img_cols || mpy >= img_rows) { free(filters); free(centers); return 1; } /* :873-885 */

    float *acc = (float *)calloc(cube, sizeof(float));
    float *filt = (float *)malloc(sizeof(float) * cube);
    float *fimg = (float *)malloc(sizeof(float) * npix * 3);
    float *dimg = fimg + npix, *gain = dimg + npix;
    for (int b = 0; b < n_filters; ++b) {
        int pr, pc;
        float wx;
        thz_oracle_band_psf(P, centers[b], dx, dy, img_rows, img_cols, NULL, &pr, &pc, &wx);
        float *psf = (float *)malloc(sizeof(float) * (size_t)pr * pc);
        thz_oracle_band_psf(P, centers[b], dx, dy, img_rows, img_cols, psf, &pr, &pc, &wx);
        thz_oracle_filter_scan(data, npix, nt, filters + (size_t)b * ntaps, ntaps, filt);
        for (size_t p = 0; p < npix; ++p) { /* mapv(x*x).sum_axis(Axis(2)), :966 */
            float s = 0.0f;
            for (int t = 0; t < nt; ++t) s += filt[p * nt + t] * filt[p * nt + t];
            fimg[p] = s;
        }
        float fi = floorf((wx - w_min) / (w_max - w_min) * ((float)n_iterations - 1.0f) + 1.0f);
        int n_iter = (fi != fi || fi < 0.0f) ? 0 : (int)fi; /* `as usize`: NaN -> 0 */
        if (n_iter_out) n_iter_out[b] = n_iter;
        thz_oracle_richardson_lucy(fimg, nx, ny, psf, pr, pc, n_iter, dimg);
        for (size_t p = 0; p < npix; ++p) {
            float uu = dimg[p] > 0.0f ? dimg[p] : 0.0f; /* x.max(0.0): NaN -> 0 */
            gain[p] = sqrtf(uu / fimg[p]);
            if (gains_out) gains_out[(size_t)b * npix + p] = gain[p];
        }
        for (size_t p = 0; p < npix; ++p)
            for (int t = 0; t < nt; ++t) acc[p * nt + t] += filt[p * nt + t] * gain[p];
        free(psf);
    }
    memcpy(out, acc, sizeof(float) * cube);
    if (img_out)
        for (size_t p = 0; p < npix; ++p) {
            float s = 0.0f;
            for (int t = 0; t < nt; ++t) s += acc[p * nt + t] * acc[p * nt + t];
            img_out[p] = s;
        }
    free(acc); free(filt); free(fimg); free(filters); free(centers);
    return 0;
}
