// deconv_host.cpp — see deconv_host.hpp.  f32/f64 operation order follows the
// reference (src/filters/deconvolution.rs:30-211, 906-960; src/filters/psf.rs);
// built with -ffp-contract=off.
#include "deconv_host.hpp"

#include <cfloat>
#include <cmath>

namespace thz {

static const float kPiF = 3.14159274101257324219f;
static const double kPi = 3.14159265358979323846;

// CubicSplineCoeffs::eval_single, psf.rs:26-80
float spline_eval(const thz_spline &s, float x)
{
    const size_t n = s.n_knots;
    if (n == 0) return 0.0f;
    if (x < s.knots[0]) {
        const float dx = x - s.knots[0];
        return std::fmax(s.coeff_a[0] + s.coeff_b[0] * dx, 1e-6f);
    }
    if (x > s.knots[n - 1]) {
        const size_t i = n - 2;
        const float de = s.knots[n - 1] - s.knots[i];
        const float y_end = s.coeff_a[i] + s.coeff_b[i] * de + s.coeff_c[i] * de * de + s.coeff_d[i] * de * de * de;
        const float slope = s.coeff_b[i] + 2.0f * s.coeff_c[i] * de + 3.0f * s.coeff_d[i] * de * de;
        return std::fmax(y_end + slope * (x - s.knots[n - 1]), 1e-6f);
    }
    size_t left = 0, right = n - 1;
    while (right - left > 1) {
        const size_t mid = (left + right) / 2;
        if (s.knots[mid] > x) right = mid; else left = mid;
    }
    const float dx = x - s.knots[left];
    return s.coeff_a[left] + s.coeff_b[left] * dx + s.coeff_c[left] * dx * dx + s.coeff_d[left] * dx * dx * dx;
}

// CubicSplineCoeffs::eval_single_const_extrap, psf.rs:83-117
float spline_eval_const(const thz_spline &s, float x)
{
    const size_t n = s.n_knots;
    if (n == 0) return 0.0f;
    if (x < s.knots[0]) return s.values[0];
    if (x > s.knots[n - 1]) return s.values[n - 1];
    size_t left = 0, right = n - 1;
    while (right - left > 1) {
        const size_t mid = (left + right) / 2;
        if (s.knots[mid] > x) right = mid; else left = mid;
    }
    const float dx = x - s.knots[left];
    return s.coeff_a[left] + s.coeff_b[left] * dx + s.coeff_c[left] * dx * dx + s.coeff_d[left] * dx * dx * dx;
}

// HybridFit::eval_single / eval_correction, psf.rs:122-179
float hybrid_eval(const thz_hybrid_fit &h, float f)
{
    const thz_spline &s = h.correction;
    const size_t n = s.n_knots;
    float corr = 0.0f;
    if (n > 0) {
        const float f_min = s.knots[0], f_max = s.knots[n - 1];
        if (f >= f_min && f <= f_max) {
            corr = spline_eval(s, f);
        } else if (f < f_min) {
            const float slope = std::fmin(s.coeff_b[0], h.base_a / (f * f));
            corr = s.coeff_a[0] + slope * (f - f_min);
        } else {
            const size_t i = n - 2;
            const float de = s.knots[n - 1] - s.knots[i];
            const float y_end = s.coeff_a[i] + s.coeff_b[i] * de + s.coeff_c[i] * de * de + s.coeff_d[i] * de * de * de;
            const float slope_end = s.coeff_b[i] + 2.0f * s.coeff_c[i] * de + 3.0f * s.coeff_d[i] * de * de;
            const float slope = std::fmin(slope_end, h.base_a / (f * f));
            corr = y_end + slope * (f - s.knots[n - 1]);
        }
    }
    const float base = h.base_a / f + h.base_b;
    return std::fmax(base + corr, 1e-6f);
}

// ---- Kaiser FIR design, deconvolution.rs:30-156 (f64) -----------------------
static double kaiser_beta_for(int ntaps, double width_ratio)
{
    double atten = 2.285 * ((double)ntaps - 1.0) * kPi * width_ratio + 7.95;
    if (atten < 0.0) atten = 0.0;
    if (atten > 50.0) return 0.1102 * (atten - 8.7);
    if (atten >= 21.0) return 0.5842 * std::pow(atten - 21.0, 0.4) + 0.07886 * (atten - 21.0);
    return 0.0;
}

static double bessel_i0(double x)
{
    double sum = 1.0, term = 1.0;
    const double q = (x / 2.0) * (x / 2.0);
    for (int k = 1; k < 50; ++k) {
        term *= q / (double)(k * k);
        sum += term;
        if (term < 1e-12 * sum) break;
    }
    return sum;
}

// Kaiser window of the design, w[n] = I0(beta sqrt(1 - arg^2)) / I0(beta) with zero end points as the
// reference writes it: it depends on (taps, beta) only, so a bank evaluates it once instead of once
// per cut-off (the Bessel series was 1.3 of the 1.5 ms a call spent designing its 25 filters).
static void kaiser_window(int adj, double beta, std::vector<double> &win)
{
    win.assign((size_t)adj, 0.0);
    for (int n = 1; n + 1 < adj; ++n) {
        const double arg = 2.0 * (double)n / ((double)adj - 1.0) - 1.0;
        win[(size_t)n] = bessel_i0(beta * std::sqrt(1.0 - arg * arg)) / bessel_i0(beta);
    }
}

static void lowpass(int n_taps, double cutoff_hz, const std::vector<double> &win, double fs, double *out)
{
    const int adj = (n_taps % 2 == 0) ? n_taps - 1 : n_taps;
    const double mid = (double)(adj - 1) / 2.0;
    const double cutoff = cutoff_hz / fs;
    double sum = 0.0;
    for (int n = 0; n < adj; ++n) {
        const double a = 2.0 * kPi * cutoff * ((double)n - mid);
        const double sinc = std::fabs(a) < 1e-10 ? 1.0 : std::sin(a) / a;
        out[n] = sinc * win[(size_t)n];
        sum += out[n];
    }
    if (std::fabs(sum) > 1e-10)
        for (int n = 0; n < adj; ++n) out[n] /= sum;
    if (n_taps % 2 == 0) out[adj] = 0.0;
}

static void highpass(int n_taps, double cutoff_hz, const std::vector<double> &win, double fs, double *out)
{
    const int adj = (n_taps % 2 == 0) ? n_taps - 1 : n_taps;
    const int mid = (int)((double)(adj - 1) / 2.0);
    lowpass(adj, cutoff_hz, win, fs, out);
    for (int i = 0; i < adj; ++i) out[i] = (i == mid) ? 1.0 - out[i] : -out[i];
    if (n_taps % 2 == 0) out[adj] = 0.0;
}

void filter_bank(int n_filters, double start_freq, double end_freq, double win_width,
                 const float *time, std::vector<float> &filters, std::vector<float> &centers)
{
    const int ntaps = kDeconvTaps;
    const double dt = (double)(time[1] - time[0]);
    const double fs = 1.0 / dt;
    const double log_start = std::log(start_freq), log_end = std::log(end_freq);
    const double log_step = (log_end - log_start) / (double)(n_filters - 1);
    centers.resize((size_t)n_filters);
    for (int i = 0; i < n_filters; ++i) centers[(size_t)i] = (float)std::exp(log_start + (double)i * log_step);
    filters.assign((size_t)n_filters * ntaps, 0.0f);
    const double beta = kaiser_beta_for(ntaps, win_width / (0.5 * fs));
    std::vector<double> h((size_t)ntaps), h2((size_t)ntaps), win;
    kaiser_window((ntaps % 2 == 0) ? ntaps - 1 : ntaps, beta, win);
    for (int i = 0; i < n_filters; ++i) {
        const double cf = (double)centers[(size_t)i];
        const double lowcut = (i == 0) ? 0.0 : std::sqrt((double)centers[(size_t)i - 1] * cf);
        const double highcut = (i == n_filters - 1) ? 0.5 * fs : std::sqrt(cf * (double)centers[(size_t)i + 1]);
        if (lowcut <= 0.0) {
            lowpass(ntaps, highcut, win, fs, h.data());
        } else if (highcut >= 0.5 * fs) {
            highpass(ntaps, lowcut, win, fs, h.data());
        } else {
            highpass(ntaps, lowcut, win, fs, h.data());
            highpass(ntaps, highcut, win, fs, h2.data());
            for (int j = 0; j < ntaps; ++j) h[(size_t)j] -= h2[(size_t)j];
        }
        for (int j = 0; j < ntaps; ++j) filters[(size_t)i * ntaps + j] = (float)h[(size_t)j];
    }
}

// ---- PSF, psf.rs:228-332 ---------------------------------------------------
static float gaussian(float xi, float x0, float w)
{
    const float d = xi - x0;
    return std::sqrt(2.0f / kPiF) * std::exp(-2.0f * (d * d) / (w * w)) / w;
}

// linear interpolation on sorted knots (stands in for the interp1d crate)
static float interp_lin(const std::vector<float> &xs, const std::vector<float> &ys, float x)
{
    const size_t n = xs.size();
    if (n == 1) return ys[0];
    size_t lo = 0, hi = n - 1;
    if (x <= xs[0]) { lo = 0; hi = 1; }
    else if (x >= xs[n - 1]) { lo = n - 2; hi = n - 1; }
    else {
        while (hi - lo > 1) {
            const size_t mid = (lo + hi) / 2;
            if (xs[mid] > x) hi = mid; else lo = mid;
        }
    }
    return ys[lo] + (ys[hi] - ys[lo]) * (x - xs[lo]) / (xs[hi] - xs[lo]);
}

static void pad_profile(std::vector<float> &x, std::vector<float> &p, float new_max)
{
    const float step = x[x.size() - 1] - x[x.size() - 2];
    const float f = std::ceil((new_max - x[x.size() - 1]) / step);
    const int extra = f > 0.0f ? (int)f : 0;
    for (int k = 0; k < extra; ++k) {
        x.push_back(x[x.size() - 1] + step);
        x.insert(x.begin(), x[0] - step);
        p.push_back(0.0f);
        p.insert(p.begin(), 0.0f);
    }
}

BandPsf band_psf(const thz_psf &P, float center_freq, float dx, float dy, int img_rows, int img_cols)
{
    BandPsf out;
    const float wx = hybrid_eval(P.wx_fit, center_freq), wy = hybrid_eval(P.wy_fit, center_freq);
    const float x0 = spline_eval_const(P.x0_spline, center_freq), y0 = spline_eval_const(P.y0_spline, center_freq);
    out.wx = wx;
    float rx = (wx + std::fabs(x0)) * 3.0f, ry = (wy + std::fabs(y0)) * 3.0f;
    if (rx < 2.5f) rx = 2.5f;
    if (ry < 2.5f) ry = 2.5f;
    rx = std::floor(rx / dx) * dx + dx;
    ry = std::floor(ry / dy) * dy + dy;
    const float cx = std::fmin(rx, ((float)img_cols - 2.0f) * dx / 2.0f);
    const float cy = std::fmin(ry, ((float)img_rows - 2.0f) * dy / 2.0f);
    const long kx = (long)std::floor(cx / dx), ky = (long)std::floor(cy / dy);
    std::vector<float> x, y, gx, gy;
    for (long i = -kx; i <= kx; ++i) { x.push_back((float)i * dx); gx.push_back(gaussian(x.back(), x0, wx)); }
    for (long i = -ky; i <= ky; ++i) { y.push_back((float)i * dy); gy.push_back(gaussian(y.back(), y0, wy)); }
    // create_psf_2d, psf.rs:228-313
    float mx = -FLT_MAX, my = -FLT_MAX, cxm = -FLT_MAX, cym = -FLT_MAX;
    for (float v : gx) mx = std::fmax(mx, v);
    for (float v : gy) my = std::fmax(my, v);
    for (float v : x) cxm = std::fmax(cxm, v);
    for (float v : y) cym = std::fmax(cym, v);
    for (float &v : gx) v /= mx;
    for (float &v : gy) v /= my;
    long x_max = (long)std::floor(cxm), y_max = (long)std::floor(cym);
    if (x_max < 0) x_max = 0;
    if (y_max < 0) y_max = 0;
    if (x.size() >= 2) pad_profile(x, gx, std::ceil(2.0f * (float)x_max));
    if (y.size() >= 2) pad_profile(y, gy, std::ceil(2.0f * (float)y_max));
    out.rows = (int)(2 * x_max + 1);
    out.cols = (int)(2 * y_max + 1);
    out.v.resize((size_t)out.rows * out.cols);
    for (long i = -x_max; i <= x_max; ++i) out.fx.push_back(interp_lin(x, gx, (float)i * dx));
    for (long j = -y_max; j <= y_max; ++j) out.fy.push_back(interp_lin(y, gy, (float)j * dy));
    for (int i = 0; i < out.rows; ++i)
        for (int j = 0; j < out.cols; ++j) out.v[(size_t)i * out.cols + (size_t)j] = out.fx[(size_t)i] * out.fy[(size_t)j];
    return out;
}

}  // namespace thz
