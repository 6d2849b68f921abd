// host_windows.cpp — O(nt) host-side multiplier vectors for the stage kernels.
//
// The reference evaluates its window formulas once per trace inside the pixel
// loops; the value only depends on the time/frequency axis, so the engine
// evaluates each formula once per call on the host, in the reference's f32
// operation order, and the kernels multiply by the resulting vector.
// Built with -ffp-contract=off (no FMA where the Rust code has mul then add).
#include "host_windows.hpp"

#include <cmath>
#include <limits>
#include <vector>

namespace thz {

static const float kPiF = 3.14159274101257324219f;  // std::f32::consts::PI

// blackman_window, math_tools.rs:81-90
static float blackman_window(float n, float m)
{
    const float res = 0.42f - 0.5f * std::cos(2.0f * kPiF * n / m) + 0.08f * std::cos(4.0f * kPiF * n / m);
    if (std::isnan(res)) return 1.0f;
    return res < 0.0f ? 0.0f : (res > 1.0f ? 1.0f : res);
}

// apply_adapted_blackman_window on a vector of ones, math_tools.rs:102-122.
// `axis` is the slice the reference passes (its own first/last coordinates are
// the taper origins).
void adapted_blackman(const float *axis, size_t len, float lower, float upper, float *out)
{
    if (len == 0) return;
    const float a0 = axis[0], an = axis[len - 1];
    for (size_t i = 0; i < len; ++i) {
        const float t = axis[i];
        float w = 1.0f;
        if (t <= lower + a0) {
            w = blackman_window(t - a0, 2.0f * lower);
        } else if (t >= an - upper) {
            w = blackman_window(t - (an - upper * 2.0f), 2.0f * upper);
        }
        out[i] = w;
    }
}

// apply_hamming / hanning / blackman / flat_top, math_tools.rs:131-198
void fft_window(int type, const float *time, size_t nt, float lower, float upper, float *out)
{
    if (type == 0) {  // AdaptedBlackman, math_tools.rs:357-364
        adapted_blackman(time, nt, lower, upper, out);
        return;
    }
    float mn = std::numeric_limits<float>::infinity(), mx = -std::numeric_limits<float>::infinity();
    for (size_t i = 0; i < nt; ++i) {
        mn = std::fmin(mn, time[i]);
        mx = std::fmax(mx, time[i]);
    }
    for (size_t i = 0; i < nt; ++i) {
        const float t = (time[i] - mn) / (mx - mn);
        float w = 1.0f;
        switch (type) {
        case 3: w = 0.54f - 0.46f * std::cos(2.0f * kPiF * t); break;                    // Hamming
        case 2: w = 0.5f * (1.0f - std::cos(2.0f * kPiF * t)); break;                    // Hanning
        case 1: w = 0.42f - 0.5f * std::cos(2.0f * kPiF * t) + 0.08f * std::cos(4.0f * kPiF * t); break;
        case 4:
            w = 1.0f - 1.93f * std::cos(2.0f * kPiF * t) + 1.29f * std::cos(4.0f * kPiF * t)
                - 0.388f * std::cos(6.0f * kPiF * t) + 0.028f * std::cos(8.0f * kPiF * t);
            break;
        default: break;
        }
        out[i] = w;
    }
}

// TimeDomainBandPass*::filter index rule, band_pass_td_before_fft.rs:134-152
void td_bandpass(const float *time, size_t nt, double *low, double *high, double width, float *out,
                 int64_t *lower_out, int64_t *upper_out)
{
    const float min_time = nt ? time[0] : 0.0f, max_time = nt ? time[nt - 1] : 0.0f;
    if (*low < (double)min_time) *low = (double)min_time;    // self.low.max(min_time)
    if (*high > (double)max_time) *high = (double)max_time;  // self.high.min(max_time)
    size_t lower = 0;
    for (size_t i = 0; i < nt; ++i)
        if (time[i] >= (float)*low) { lower = i; break; }
    size_t upper = nt ? nt - 1 : 0;
    for (size_t i = 0; i < nt; ++i)
        if (time[i] >= (float)*high) { upper = i; break; }
    if (upper < lower + 1) upper = lower + 1;
    if (upper > nt) upper = nt;
    for (size_t i = 0; i < nt; ++i) out[i] = 0.0f;
    if (upper > lower) adapted_blackman(time + lower, upper - lower, (float)width, (float)width, out + lower);
    if (lower_out) *lower_out = (int64_t)lower;
    if (upper_out) *upper_out = (int64_t)upper;
}

// FrequencyDomainBandPass::filter index rule + taper, band_pass_fd.rs:134-168
void fd_bandpass(const float *freq, size_t nf, double low, double high, double width, float *out,
                 int64_t *lower_out, int64_t *upper_out)
{
    const float safe_low = (float)(low > 0.0 ? low : 0.0);
    const double last = nf ? (double)freq[nf - 1] : 10.0;
    const float safe_high = (float)(high < last ? high : last);
    size_t lower = 0, upper = nf;
    for (size_t i = 0; i < nf; ++i)
        if (freq[i] >= safe_low) { lower = i; break; }
    for (size_t i = nf; i-- > 0;)
        if (freq[i] <= safe_high) { upper = i + 1; break; }
    for (size_t i = 0; i < nf; ++i) out[i] = 0.0f;
    if (upper > lower) adapted_blackman(freq + lower, upper - lower, (float)width, (float)width, out + lower);
    if (lower_out) *lower_out = (int64_t)lower;
    if (upper_out) *upper_out = (int64_t)upper;
}

// ---- build-defined frequency-domain multipliers (not in the reference)

// K14 water-vapour line notch: prod_i (1 - exp(-((f - f_i)/sigma)^2)), f32
void water_line_mask(const float *freq, size_t nf, const float *lines, size_t n_lines, float sigma,
                     float *out)
{
    // evaluated in double and rounded once: the notch flanks are steep ((f - line) / sigma amplifies the f32
    // rounding of the axis a hundredfold), and this O(nf x lines) host loop costs nothing
    for (size_t k = 0; k < nf; ++k) {
        double m = 1.0;
        for (size_t i = 0; i < n_lines; ++i) {
            const double z = ((double)freq[k] - (double)lines[i]) / (double)sigma;
            m *= 1.0 - std::exp(-(z * z));
        }
        out[k] = (float)m;
    }
}

// K13 reference-pulse Wiener deconvolution: H = conj(R) / (|R|^2 + eps_rel * max|R|^2)
void wiener_filter(const float *ref_fft, size_t nf, float eps_rel, float *out)
{
    float mx = 0.0f;
    for (size_t k = 0; k < nf; ++k) {
        const float p = ref_fft[2 * k] * ref_fft[2 * k] + ref_fft[2 * k + 1] * ref_fft[2 * k + 1];
        if (p > mx) mx = p;
    }
    const float eps = eps_rel * mx;
    for (size_t k = 0; k < nf; ++k) {
        const float re = ref_fft[2 * k], im = ref_fft[2 * k + 1];
        const float den = re * re + im * im + eps;
        out[2 * k] = den > 0.0f ? re / den : 0.0f;
        out[2 * k + 1] = den > 0.0f ? -im / den : 0.0f;
    }
}

// ndarray::Array1::linspace: start + step * i
static void linspace(float a, float b, size_t n, float *out)
{
    const float step = n > 1 ? (b - a) / (float)(n - 1) : 0.0f;
    for (size_t i = 0; i < n; ++i) out[i] = a + step * (float)i;
}

// TiltCompensation::filter geometry, tilt_compensation.rs:104-175: time extension
// and the per-pixel index at which the trace is inserted into the extended axis.
size_t tilt_plan(const float *time, size_t nt, size_t nx, size_t ny, double tilt_x_deg,
                 double tilt_y_deg, float dx, float dy, float *new_time, int32_t *insert_index)
{
    const float time_shift_x = (float)tilt_x_deg / 180.0f * kPiF;
    const float time_shift_y = (float)tilt_y_deg / 180.0f * kPiF;
    const float center_x = (float)nx / 2.0f * dx;
    const float center_y = (float)ny / 2.0f * dy;
    const double c = 0.299792458;  // mm/ps
    const float dt = 0.05f;        // hard-coded in the reference (:122)
    const float max_offset_x = (float)((double)center_x * (double)std::fabs(time_shift_x) / c);
    const float max_offset_y = (float)((double)center_y * (double)std::fabs(time_shift_y) / c);
    float extension = (max_offset_x + max_offset_y) / dt;
    extension = std::floor(extension) * dt;
    const size_t num_steps = (size_t)std::round(extension / dt);
    const size_t ext = nt + 2 * num_steps;
    if (new_time && nt) {
        linspace(time[0] - extension, time[0] - dt, num_steps, new_time);
        for (size_t i = 0; i < nt; ++i) new_time[num_steps + i] = time[i];
        linspace(time[nt - 1] + dt, time[nt - 1] + extension, num_steps, new_time + num_steps + nt);
    }
    if (insert_index) {
        for (size_t i = 0; i < nx; ++i)
            for (size_t j = 0; j < ny; ++j) {
                const float x_offset =
                    (float)((double)(((float)i - (float)nx / 2.0f) * dx) * (double)time_shift_x / c);
                const float y_offset =
                    (float)((double)(((float)j - (float)ny / 2.0f) * dy) * (double)time_shift_y / c);
                const float delta = x_offset + y_offset;
                const long delta_steps = (long)std::floor(delta / dt);
                long ins = (long)num_steps + delta_steps;
                if (ins < 0) ins = 0;
                if (ins > (long)ext) ins = (long)ext;
                insert_index[i * ny + j] = (int32_t)ins;
            }
    }
    return num_steps;
}

// ConfigCommand::OpenRef, data_thread.rs:405-481: zero-padded index-shift alignment of a
// reference pulse to the scan's time axis.  Returns 0 untouched, 1 shifted, 2 naive resize.
int align_reference(const float *scan_time, size_t nt, const float *ref_time, const float *ref_signal, size_t nref,
                    float *out)
{
    if (!(nt != nref || (nref > 0 && std::fabs(scan_time[0] - ref_time[0]) > 1e-9f))) {
        for (size_t i = 0; i < nt; ++i) out[i] = ref_signal[i];
        return 0;
    }
    if (nt > 1 && nref > 1) {
        for (size_t i = 0; i < nt; ++i) out[i] = 0.0f;
        const float ref_dt = ref_time[1] - ref_time[0];
        const float time_offset = scan_time[0] - ref_time[0];
        const float q = std::round(time_offset / ref_dt);
        // Rust `as isize`: NaN -> 0, saturating at the ends
        long long index_offset = 0;
        if (q >= 9.2e18f) index_offset = std::numeric_limits<long long>::max();
        else if (q <= -9.2e18f) index_offset = std::numeric_limits<long long>::min();
        else if (q == q) index_offset = (long long)q;
        const size_t src_start = index_offset > 0 ? (size_t)index_offset : 0;
        const size_t dst_start = index_offset < 0 ? (size_t)(-(index_offset + 1)) + 1 : 0;
        const size_t src_len = nref > src_start ? nref - src_start : 0;
        const size_t dst_len = nt > dst_start ? nt - dst_start : 0;
        const size_t copy_len = src_len < dst_len ? src_len : dst_len;
        for (size_t i = 0; i < copy_len; ++i) out[dst_start + i] = ref_signal[src_start + i];
        return 1;
    }
    for (size_t i = 0; i < nt; ++i) out[i] = i < nref ? ref_signal[i] : 0.0f;
    return 2;
}

// Window step of OpenRef (:490-515): the multiplier comes from the reference file's own time
// axis; the adapted Blackman stops at the shorter of signal and axis (iter().zip()), the
// other windows need equal lengths (ndarray Zip panics otherwise) -> false.
bool reference_window(int type, const float *ref_time, size_t nref, float lower, float upper, size_t nt, float *win)
{
    for (size_t i = 0; i < nt; ++i) win[i] = 1.0f;
    if (type != 0 && nref != nt) return false;
    if (nref == 0) return true;
    std::vector<float> w(nref);
    fft_window(type, ref_time, nref, lower, upper, w.data());
    for (size_t i = 0; i < nt && i < nref; ++i) win[i] = w[i];
    return true;
}

// calculate_optical_properties, math_tools.rs:663-701 (f32, the reference's operation order)
void optical_properties(const float *sample_amp, const float *sample_phase, const float *ref_amp,
                        const float *ref_phase, const float *freq, size_t nf, float thickness, float *n_out,
                        float *alpha_out, float *kappa_out)
{
    const float c = 2.99792458e8f;
    for (size_t i = 0; i < nf; ++i) {
        const float frequency_hz = freq[i] * 1.0e12f;
        const float delta_phi = sample_phase[i] - ref_phase[i];
        const float omega = 2.0f * kPiF * frequency_hz;
        const float n = 1.0f + c * delta_phi / (omega * thickness);
        const float amp = std::fmax(sample_amp[i], 1e-12f);
        const float amp_ref = std::fmax(ref_amp[i], 1e-12f);
        const float n_safe = std::fmax(n, 1e-6f);
        const float alpha =
            -2.0f / thickness * std::log(((n_safe + 1.0f) * (n_safe + 1.0f)) / (4.0f * n_safe) * amp / amp_ref);
        n_out[i] = n;
        alpha_out[i] = alpha;
        kappa_out[i] = alpha * c / (4.0f * kPiF * frequency_hz);
    }
}

// ---- 3-D voxel envelope, gui/threed_plot.rs

// gaussian_kernel1d, threed_plot.rs:80-101
void gaussian_kernel1d(float sigma, int radius, float *out)
{
    const int size = 2 * radius + 1;
    const float sigma2 = 2.0f * sigma * sigma;
    float sum = 0.0f;
    for (int i = 0; i < size; ++i) {
        const float x = (float)i - (float)radius;
        const float value = std::exp(-x * x / sigma2);
        sum += value;
        out[i] = value;
    }
    for (int i = 0; i < size; ++i) out[i] /= sum;
}

// cube size, spacing and half extents of instance_from_data, threed_plot.rs:147-160, 225-231
VoxelLayout voxel_layout(float time_span, size_t gw, size_t gh, size_t gd, size_t ow, size_t oh, size_t od)
{
    VoxelLayout L;
    const float base = 1.0f / 4.0f;
    const float c = 300000000.0f;
    L.cube_width = base;
    L.cube_height = base;
    L.cube_depth = base / (time_span * c / 1.0e9f * 2.0f);
    L.spacing_w = ((float)ow * L.cube_width) / (float)gw;
    L.spacing_h = ((float)oh * L.cube_height) / (float)gh;
    L.spacing_d = ((float)od * L.cube_depth) / (float)gd;
    L.half_w = ((float)ow * base) / 2.0f;
    L.half_h = ((float)oh * base) / 2.0f;
    L.half_d = ((float)od * L.cube_depth) / 2.0f;
    return L;
}

// One level of the radix select for the k-th largest value (k >= 1): walks the
// histogram from the top bin down; *bin holds the value, *k_rem its rank inside
// that bin.  Returns 0, or -1 when the histogram holds fewer than k values.
int select_step(const unsigned long long *hist, int nbins, unsigned long long k, int *bin,
                unsigned long long *k_rem)
{
    unsigned long long above = 0;
    for (int b = nbins - 1; b >= 0; --b) {
        if (above + hist[b] >= k) {
            *bin = b;
            *k_rem = k - above;
            return 0;
        }
        above += hist[b];
    }
    return -1;
}

}  // namespace thz
