// api.cpp — the C ABI of libthzgpu.so (include/thzgpu.h).
//
// Thin: argument checks, plan/table ownership, launches on the context's
// stream.  No CPU compute path exists here — without a HIP device every entry
// point that would compute returns THZ_ERR_HIP.
#include "../../include/thzgpu.h"

#include "host_windows.hpp"
#include "kernels.hpp"
#include "plan_host.hpp"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace thz;

#include "ctx.hpp"

extern "C" {

int thz_abi_version(void) { return THZGPU_ABI_VERSION; }

int thz_create(int device, thz_ctx **out)
{
    if (!out) return THZ_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return THZ_ERR_HIP;
    thz_ctx *ctx = new thz_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        delete ctx;
        return THZ_ERR_HIP;
    }
    *out = ctx;
    return THZ_OK;
}

void thz_destroy(thz_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->d_tables) (void)hipFree(ctx->d_tables);
    if (ctx->d_big) (void)hipFree(ctx->d_big);
    if (ctx->ws) (void)hipFree(ctx->ws);
    for (auto &g : ctx->dc_graph) g.drop();
    ctx->drop_dc_tables();
    for (auto &b : ctx->dc_pool) (void)hipFree(b.p);
    for (auto &r : ctx->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : ctx->pool) (void)hipEventDestroy(e);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    for (hipStream_t st : ctx->aux_streams)
        if (st) (void)hipStreamDestroy(st);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int thz_release_scratch(thz_ctx *ctx)
{
    if (!ctx) return THZ_ERR_INVALID;
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (hipStream_t st : ctx->aux_streams)
        if (st) HIP_TRY(ctx, hipStreamSynchronize(st));
    for (auto &g : ctx->dc_graph) g.drop();
    ctx->drop_dc_tables();
    for (auto &b : ctx->dc_pool) (void)hipFree(b.p);
    ctx->dc_pool.clear();
    if (ctx->ws) (void)hipFree(ctx->ws);
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
    return THZ_OK;
}

const char *thz_last_error(const thz_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

void *thz_stream(thz_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int thz_sync(thz_ctx *ctx)
{
    if (!ctx) return THZ_ERR_INVALID;
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return THZ_OK;
}

int thz_malloc(thz_ctx *ctx, void **d_ptr, size_t bytes)
{
    if (!ctx || !d_ptr) return THZ_ERR_INVALID;
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(ctx, hipMalloc(d_ptr, bytes ? bytes : 1));
    return THZ_OK;
}

int thz_free(thz_ctx *ctx, void *d_ptr)
{
    if (!ctx) return THZ_ERR_INVALID;
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipFree(d_ptr));
    return THZ_OK;
}

int thz_memcpy_h2d(thz_ctx *ctx, void *d_dst, const void *src, size_t bytes)
{
    if (!ctx || (bytes && (!d_dst || !src))) return THZ_ERR_INVALID;
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return THZ_OK;
}

int thz_memcpy_d2h(thz_ctx *ctx, void *dst, const void *d_src, size_t bytes)
{
    if (!ctx || (bytes && (!dst || !d_src))) return THZ_ERR_INVALID;
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return THZ_OK;
}

int thz_memcpy_d2d(thz_ctx *ctx, void *d_dst, const void *d_src, size_t bytes)
{
    if (!ctx || (bytes && (!d_dst || !d_src))) return THZ_ERR_INVALID;
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return THZ_OK;
}

int thz_memset(thz_ctx *ctx, void *d_dst, int value, size_t bytes)
{
    if (!ctx || (bytes && !d_dst)) return THZ_ERR_INVALID;
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(ctx, hipMemsetAsync(d_dst, value, bytes, ctx->stream));
    return THZ_OK;
}

/* ---------------------------------------------------------------- plan */

int thz_set_time_axis(thz_ctx *ctx, const float *time, size_t nt)
{
    if (!ctx || !time || nt < 2) return fail(ctx, THZ_ERR_INVALID, "time axis needs >= 2 samples");
    if (int rc = use_device(ctx)) return rc;
    // the same axis under the same kernel-family switches: the plan and its device tables stand (a session
    // re-plans at every recompute, data_thread.rs:1194-1227 — tables for nt = 4096 are thousands of double-precision
    // sines, an allocation and a blocking upload: 0.1 ms that a 2 ms slab recompute would feel)
    if (ctx->have_plan && ctx->time.size() == nt && ctx->plan_allow_f == ctx->allow_f && ctx->plan_allow_p == ctx->allow_p
        && std::memcmp(ctx->time.data(), time, nt * sizeof(float)) == 0)
        return THZ_OK;
    PlanHost H;
    if (!build_plan(nt, H, ctx->allow_f, ctx->allow_p))
        return fail(ctx, THZ_ERR_UNSUPPORTED,
                    "unsupported trace length " + std::to_string(nt) +
                        " (any length 2..65536)");
    const size_t n_tw = H.tw.size(), n_sp = H.tw_split.size(), n_ch = H.chirp_conj.size(),
                 n_bf = H.bfft.size();
    const size_t n_f1 = H.f_t1.size(), n_f2 = H.f_t2.size(), n_fw = H.f_w2n.size();
    const size_t n_ones = (size_t)(H.nf + 1) / 2;  // nf floats of 1.0, counted in c32 units
    const size_t n_p1 = H.p_t1.size(), n_p2 = H.p_t2.size();
    const size_t total = n_tw + n_sp + n_ch + n_bf + n_f1 + n_f2 + n_fw + n_ones + n_p1 + n_p2;
    c32 *d = nullptr;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMalloc((void **)&d, total * sizeof(c32)));
    std::vector<c32> pack;
    pack.reserve(total);
    pack.insert(pack.end(), H.tw.begin(), H.tw.end());
    pack.insert(pack.end(), H.tw_split.begin(), H.tw_split.end());
    pack.insert(pack.end(), H.chirp_conj.begin(), H.chirp_conj.end());
    pack.insert(pack.end(), H.bfft.begin(), H.bfft.end());
    pack.insert(pack.end(), H.f_t1.begin(), H.f_t1.end());
    pack.insert(pack.end(), H.f_t2.begin(), H.f_t2.end());
    pack.insert(pack.end(), H.f_w2n.begin(), H.f_w2n.end());
    pack.insert(pack.end(), n_ones, c32{1.0f, 1.0f});
    pack.insert(pack.end(), H.p_t1.begin(), H.p_t1.end());
    pack.insert(pack.end(), H.p_t2.begin(), H.p_t2.end());
    hipError_t e = hipMemcpy(d, pack.data(), total * sizeof(c32), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        return fail(ctx, THZ_ERR_HIP, std::string("table upload: ") + hipGetErrorString(e));
    }
    c32 *d_big = nullptr;
    if (H.big && hipMalloc((void **)&d_big, (size_t)H.big_waves * (size_t)H.lds_per_wave) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(d);
        return fail(ctx, THZ_ERR_HIP, "scratch of the long-trace transform: allocation failed");
    }
    if (ctx->d_tables) (void)hipFree(ctx->d_tables);
    if (ctx->d_big) (void)hipFree(ctx->d_big);
    ctx->d_tables = d;
    ctx->d_big = d_big;
    ctx->plan_h = H;
    const size_t o_f = n_tw + n_sp + n_ch + n_bf;
    ctx->plan_d = plan_dev(H, d, n_sp ? d + n_tw : nullptr, n_ch ? d + n_tw + n_sp : nullptr,
                           n_bf ? d + n_tw + n_sp + n_ch : nullptr, n_f1 ? d + o_f : nullptr,
                           n_f2 ? d + o_f + n_f1 : nullptr, n_fw ? d + o_f + n_f1 + n_f2 : nullptr,
                           reinterpret_cast<const float *>(d + o_f + n_f1 + n_f2 + n_fw),
                           n_p1 ? d + o_f + n_f1 + n_f2 + n_fw + n_ones : nullptr,
                           n_p2 ? d + o_f + n_f1 + n_f2 + n_fw + n_ones + n_p1 : nullptr, d_big);
    ctx->plan_allow_f = ctx->allow_f;
    ctx->plan_allow_p = ctx->allow_p;
    ctx->time.assign(time, time + nt);
    ctx->freq.resize(nt / 2 + 1);
    (void)thz_host_frequency_axis(time, nt, ctx->freq.data());
    ctx->have_plan = true;
    return THZ_OK;
}

int thz_set_kernel_family(thz_ctx *ctx, int family)
{
    if (!ctx || family < 0 || family > 2) return THZ_ERR_INVALID;
    ctx->allow_f = family != 1;
    ctx->allow_p = family == 0;
    if (ctx->have_plan) {
        std::vector<float> t = ctx->time;
        return thz_set_time_axis(ctx, t.data(), t.size());
    }
    return THZ_OK;
}

size_t thz_nt(const thz_ctx *ctx) { return ctx && ctx->have_plan ? ctx->time.size() : 0; }
size_t thz_nf(const thz_ctx *ctx) { return ctx && ctx->have_plan ? ctx->freq.size() : 0; }

int thz_get_frequency(const thz_ctx *ctx, float *frequency)
{
    if (!ctx || !frequency) return THZ_ERR_INVALID;
    if (!ctx->have_plan) return THZ_ERR_NOT_READY;
    std::memcpy(frequency, ctx->freq.data(), ctx->freq.size() * sizeof(float));
    return THZ_OK;
}

const char *thz_kernel_variant(const thz_ctx *ctx)
{
    return (ctx && ctx->have_plan) ? ctx->plan_h.variant : "";
}

/* ------------------------------------------------------ host multipliers */

int thz_host_frequency_axis(const float *time, size_t nt, float *frequency)
{
    if (!time || !frequency || nt < 2) return THZ_ERR_INVALID;
    const float rng = time[nt - 1] - time[0];
    for (size_t i = 0; i < nt / 2 + 1; ++i) frequency[i] = (float)i / rng;
    return THZ_OK;
}

int thz_host_fft_window(const float *time, size_t nt, const thz_window_cfg *cfg, float *out)
{
    if (!time || !cfg || !out || cfg->type < 0 || cfg->type > 4) return THZ_ERR_INVALID;
    fft_window(cfg->type, time, nt, cfg->lower, cfg->upper, out);
    return THZ_OK;
}

int thz_host_adapted_blackman(const float *axis, size_t len, float lower, float upper, float *out)
{
    if (!axis || !out) return THZ_ERR_INVALID;
    adapted_blackman(axis, len, lower, upper, out);
    return THZ_OK;
}

int thz_host_td_bandpass(const float *time, size_t nt, double *low, double *high,
                         double window_width, float *out, int64_t *lower, int64_t *upper)
{
    if (!time || !low || !high || !out) return THZ_ERR_INVALID;
    td_bandpass(time, nt, low, high, window_width, out, lower, upper);
    return THZ_OK;
}

int thz_host_fd_bandpass(const float *frequency, size_t nf, double low, double high,
                         double window_width, float *out, int64_t *lower, int64_t *upper)
{
    if (!frequency || !out) return THZ_ERR_INVALID;
    fd_bandpass(frequency, nf, low, high, window_width, out, lower, upper);
    return THZ_OK;
}

int thz_host_water_line_mask(const float *frequency, size_t nf, const float *lines_thz,
                             size_t n_lines, float sigma_thz, float *out)
{
    if (!frequency || !out || (n_lines && !lines_thz) || !(sigma_thz > 0.0f)) return THZ_ERR_INVALID;
    water_line_mask(frequency, nf, lines_thz, n_lines, sigma_thz, out);
    return THZ_OK;
}

int thz_host_wiener_filter(const float *ref_fft, size_t nf, float eps_rel, float *out_cmask)
{
    if (!ref_fft || !out_cmask || !(eps_rel >= 0.0f)) return THZ_ERR_INVALID;
    wiener_filter(ref_fft, nf, eps_rel, out_cmask);
    return THZ_OK;
}

size_t thz_host_tilt_plan(const float *time, size_t nt, size_t nx, size_t ny, double tilt_x_deg,
                          double tilt_y_deg, float dx, float dy, float *new_time,
                          int32_t *insert_index)
{
    if (!time || nt == 0) return 0;
    return tilt_plan(time, nt, nx, ny, tilt_x_deg, tilt_y_deg, dx, dy, new_time, insert_index);
}

int thz_host_align_reference(const float *scan_time, size_t nt, const float *ref_time, const float *ref_signal,
                             size_t nref, float *out)
{
    if (!scan_time || !out || nt == 0 || (nref && (!ref_time || !ref_signal))) return THZ_ERR_INVALID;
    return align_reference(scan_time, nt, ref_time, ref_signal, nref, out);
}

int thz_reference_spectrum(thz_ctx *ctx, const float *scan_time, size_t nt, const float *ref_time,
                           const float *ref_signal, size_t nref, const thz_window_cfg *window,
                           float *reference_out, float *amplitudes, float *phases)
{
    if (!ctx) return THZ_ERR_INVALID;
    if (!scan_time || nt < 2 || !window || !reference_out || !amplitudes || !phases
        || (nref && (!ref_time || !ref_signal)))
        return fail(ctx, THZ_ERR_INVALID, "thz_reference_spectrum: bad argument");
    if (int rc = use_device(ctx)) return rc;
    std::vector<float> win(nt);
    if (!reference_window(window->type, ref_time, nref, window->lower, window->upper, nt, win.data()))
        return fail(ctx, THZ_ERR_INVALID,
                    "thz_reference_spectrum: this window needs a reference of the scan's length (the reference panics)");
    align_reference(scan_time, nt, ref_time, ref_signal, nref, reference_out);
    for (size_t i = 0; i < nt; ++i) reference_out[i] *= win[i];
    if (!ctx->have_plan || ctx->time.size() != nt || std::memcmp(ctx->time.data(), scan_time, nt * sizeof(float)) != 0)
        if (int rc = thz_set_time_axis(ctx, scan_time, nt)) return rc;
    const size_t nf = nt / 2 + 1;
    float *d = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&d, (nt + 2 * nf) * sizeof(float)));
    int rc = thz_memcpy_h2d(ctx, d, reference_out, nt * sizeof(float));
    if (!rc) rc = thz_fft(ctx, 1, d, nullptr, nullptr, nullptr, nullptr, d + nt, d + nt + nf, nullptr);
    if (!rc) rc = thz_memcpy_d2h(ctx, amplitudes, d + nt, nf * sizeof(float));
    if (!rc) rc = thz_memcpy_d2h(ctx, phases, d + nt + nf, nf * sizeof(float));
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    return rc;
}

int thz_host_optical_properties(const float *sample_amp, const float *sample_phase, const float *ref_amp,
                                const float *ref_phase, const float *freq, size_t nf, float thickness,
                                float *refractive_index, float *absorption_coeff, float *extinction_coeff)
{
    if (!sample_amp || !sample_phase || !ref_amp || !ref_phase || !freq || !refractive_index || !absorption_coeff
        || !extinction_coeff)
        return THZ_ERR_INVALID;
    optical_properties(sample_amp, sample_phase, ref_amp, ref_phase, freq, nf, thickness, refractive_index,
                       absorption_coeff, extinction_coeff);
    return THZ_OK;
}

/* ------------------------------------------------------------ stages */

int thz_fft(thz_ctx *ctx, size_t npix, const float *d_in, const float *d_win_a,
            const float *d_win_b, float *d_data_out, float *d_fft, float *d_amp, float *d_phase,
            const float *d_fd_mask)
{
    if (int rc = need_plan(ctx)) return rc;
    if (!d_in) return fail(ctx, THZ_ERR_INVALID, "thz_fft: d_in is null");
    if (npix == 0) return THZ_OK;
    StageTimer t(ctx, THZ_STAGE_FFT);
    launch_fft_fwd(ctx->stream, ctx->plan_d, npix, d_in, d_win_a, d_win_b, d_data_out,
                   reinterpret_cast<c32 *>(d_fft), d_amp, d_phase, d_fd_mask);
    return check_launch(ctx);
}

int thz_apply_fd_mask(thz_ctx *ctx, size_t npix, float *d_fft, float *d_amp, const float *d_mask)
{
    if (int rc = need_plan(ctx)) return rc;
    if (!d_mask) return fail(ctx, THZ_ERR_INVALID, "thz_apply_fd_mask: d_mask is null");
    if (npix == 0 || (!d_fft && !d_amp)) return THZ_OK;
    StageTimer t(ctx, THZ_STAGE_FD_MASK);
    launch_fd_mask(ctx->stream, npix, ctx->plan_d.nf, reinterpret_cast<c32 *>(d_fft), d_amp, d_mask);
    return check_launch(ctx);
}

int thz_apply_fd_cmask(thz_ctx *ctx, size_t npix, float *d_fft, float *d_amp, const float *d_cmask)
{
    if (int rc = need_plan(ctx)) return rc;
    if (!d_cmask) return fail(ctx, THZ_ERR_INVALID, "thz_apply_fd_cmask: d_cmask is null");
    if (npix == 0 || (!d_fft && !d_amp)) return THZ_OK;
    StageTimer t(ctx, THZ_STAGE_FD_MASK);
    launch_fd_cmask(ctx->stream, npix, ctx->plan_d.nf, ctx->plan_d.nt, reinterpret_cast<c32 *>(d_fft),
                    d_amp, reinterpret_cast<const c32 *>(d_cmask));
    return check_launch(ctx);
}

int thz_ifft(thz_ctx *ctx, size_t npix, const float *d_fft, const float *d_td_win,
             float *d_data_out, float *d_img)
{
    if (int rc = need_plan(ctx)) return rc;
    if (!d_fft || !d_data_out) return fail(ctx, THZ_ERR_INVALID, "thz_ifft: null input/output");
    if (npix == 0) return THZ_OK;
    StageTimer t(ctx, THZ_STAGE_IFFT);
    launch_fft_inv(ctx->stream, ctx->plan_d, npix, reinterpret_cast<const c32 *>(d_fft), d_td_win,
                   d_data_out, d_img);
    return check_launch(ctx);
}

int thz_polar_ifft(thz_ctx *ctx, const float *amp, const float *phase, int zero_dc_imag, float *out)
{
    if (int rc = need_plan(ctx)) return rc;
    if (!amp || !phase || !out) return fail(ctx, THZ_ERR_INVALID, "thz_polar_ifft: null argument");
    const size_t nt = ctx->time.size(), nf = nt / 2 + 1;
    std::vector<float> spec(2 * nf);
    for (size_t k = 0; k < nf; ++k) {  // Complex::from_polar, f32
        spec[2 * k] = amp[k] * std::cos(phase[k]);
        spec[2 * k + 1] = amp[k] * std::sin(phase[k]);
    }
    if (zero_dc_imag) spec[1] = 0.0f;
    float *d = nullptr;
    const size_t o = (2 * nf + 3) & ~(size_t)3;  // the trace starts 16-byte aligned, like a cube's first row
    HIP_TRY(ctx, hipMalloc((void **)&d, (o + nt) * sizeof(float)));
    int rc = thz_memcpy_h2d(ctx, d, spec.data(), 2 * nf * sizeof(float));
    if (!rc) rc = thz_ifft(ctx, 1, d, nullptr, d + o, nullptr);
    if (!rc) rc = thz_memcpy_d2h(ctx, out, d + o, nt * sizeof(float));
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    return rc;
}

int thz_pipeline(thz_ctx *ctx, size_t npix, const float *d_raw, const float *d_pre_win,
                 const float *d_fd_mask, const float *d_post_win, float *d_fft, float *d_amp,
                 float *d_phase, float *d_data_out, float *d_img)
{
    if (int rc = need_plan(ctx)) return rc;
    if (!d_raw || !d_data_out) return fail(ctx, THZ_ERR_INVALID, "thz_pipeline: null input/output");
    if (npix == 0) return THZ_OK;
    if (ctx->plan_d.big_scratch && !d_fft) return fail(ctx, THZ_ERR_INVALID, "thz_pipeline: d_fft is required for this trace length");
    StageTimer t(ctx, THZ_STAGE_PIPELINE);
    if (ctx->plan_d.mode == kModePow2
        || ((ctx->plan_d.family == kFamilyFB || ctx->plan_d.family == kFamilyFB2 || ctx->plan_d.family == kFamilyFB4
             || ctx->plan_d.family == kFamilyFB8 || ctx->plan_d.family == kFamilyP)
            && d_fft && d_amp && d_phase)) {  // mixed radix / chirp-z over the F core: one launch
        launch_pipeline(ctx->stream, ctx->plan_d, npix, d_raw, d_pre_win, d_fd_mask, d_post_win,
                        reinterpret_cast<c32 *>(d_fft), d_amp, d_phase, d_data_out, d_img);
        return check_launch(ctx);
    }
    // chirp-z lengths: forward and inverse launches; the spectrum has to be
    // materialised between them
    if (!d_fft) return fail(ctx, THZ_ERR_INVALID, "thz_pipeline: d_fft is required for this trace length");
    launch_fft_fwd(ctx->stream, ctx->plan_d, npix, d_raw, d_pre_win, nullptr, nullptr,
                   reinterpret_cast<c32 *>(d_fft), d_amp, d_phase, d_fd_mask);
    if (int rc = check_launch(ctx)) return rc;
    launch_fft_inv(ctx->stream, ctx->plan_d, npix, reinterpret_cast<const c32 *>(d_fft), d_post_win,
                   d_data_out, d_img);
    return check_launch(ctx);
}

int thz_pipeline_ex(thz_ctx *ctx, size_t npix, const thz_pipeline_io *io)
{
    return pipeline_ex_band(ctx, npix, io, io ? io->band_lo : 0, io ? io->band_hi : 0);
}

}  // extern "C"

// thz_pipeline_ex for a caller that KNOWS where its real multiplier is not zero (the session: thz_host_fd_bandpass hands
// the band's first and last bin back): every bin outside [band_lo, band_hi) is zero in io->d_fd_mask.  With a complex
// multiplier the nt = 4096 kernel then stages only the band's bins (fft_f.hpp, kCfgBand).  0, 0: unknown.
int pipeline_ex_band(thz_ctx *ctx, size_t npix, const thz_pipeline_io *io, size_t band_lo, size_t band_hi)
{
    if (int rc = need_plan(ctx)) return rc;
    int band_lo4 = 0, band_n = 0;
    if (band_hi > band_lo && io && io->d_fd_cmask) {
        band_lo4 = (int)(band_lo & ~(size_t)3);
        band_n = (int)(((band_hi + 3) & ~(size_t)3) - (size_t)band_lo4);
    }
    if (!io || !io->d_raw || !io->d_data_out || !io->d_fft || !io->d_amp || !io->d_phase)
        return fail(ctx, THZ_ERR_INVALID, "thz_pipeline_ex: d_raw, d_fft, d_amp, d_phase and d_data_out are required");
    const size_t nf = (size_t)ctx->plan_d.nf;
    if (npix == 0) {
        if (io->d_sums) HIP_TRY(ctx, hipMemsetAsync(io->d_sums, 0, 2 * nf * sizeof(float), ctx->stream));
        return THZ_OK;
    }
    // Pixel sums: inside the launch where the plan's fused kernel can (the F kernels, nt = 1024 / 2048 / 4096: the
    // block's waves add their amplitudes and unwrapped phases to one set of accumulators in LDS, group by group in
    // ticket order — fft_f.hpp, FSums — and a small pass adds the blocks' rows: 15.4 against 17.8 ms per Mi traces of
    // 4096 samples, profiles/r02_sums_in_kernel.txt), otherwise as a second pass over the two arrays just written
    // (8 nf bytes per trace).  THZ_NO_FUSED_SUMS: developer knob, forces the second pass for A/B measurements.
    const size_t sum_rows = (io->d_sums && !getenv("THZ_NO_FUSED_SUMS"))
                                ? pipeline_sum_rows(ctx->plan_d, npix, io->d_fd_cmask != nullptr, band_lo4, band_n) : 0;
    float *d_partial = nullptr;
    if (sum_rows) {
        if (int rc = ensure_ws(ctx, sum_rows * 2 * nf * sizeof(float))) return rc;
        d_partial = reinterpret_cast<float *>(ctx->ws);
    }
    {
        StageTimer t(ctx, THZ_STAGE_PIPELINE);
        launch_pipeline(ctx->stream, ctx->plan_d, npix, io->d_raw, io->d_pre_win, io->d_fd_mask, io->d_post_win,
                        reinterpret_cast<c32 *>(io->d_fft), io->d_amp, io->d_phase, io->d_data_out, io->d_img,
                        reinterpret_cast<const c32 *>(io->d_fd_cmask), d_partial, band_lo4, band_n);
        if (int rc = check_launch(ctx)) return rc;
    }
    if (!io->d_sums) return THZ_OK;
    if (d_partial) {
        StageTimer t(ctx, THZ_STAGE_MEAN);
        launch_sum_rows_f64(ctx->stream, d_partial, sum_rows, 2 * nf, io->d_sums);
        return check_launch(ctx);
    }
    if (int rc = thz_pixel_sum(ctx, npix, nf, 1, io->d_amp, io->d_sums)) return rc;
    return thz_pixel_sum(ctx, npix, nf, 1, io->d_phase, io->d_sums + nf);
}

extern "C" {

int thz_apply_td_window(thz_ctx *ctx, size_t npix, const float *d_in, const float *d_win,
                        float *d_out)
{
    if (int rc = need_plan(ctx)) return rc;
    if (!d_in || !d_win || !d_out) return fail(ctx, THZ_ERR_INVALID, "thz_apply_td_window: null pointer");
    if (npix == 0) return THZ_OK;
    StageTimer t(ctx, THZ_STAGE_TD_WINDOW);
    launch_td_window(ctx->stream, npix, ctx->plan_d.nt, d_in, d_win, d_out);
    return check_launch(ctx);
}

int thz_intensity(thz_ctx *ctx, size_t npix, const float *d_data, float *d_img)
{
    if (int rc = need_plan(ctx)) return rc;
    if (!d_data || !d_img) return fail(ctx, THZ_ERR_INVALID, "thz_intensity: null pointer");
    if (npix == 0) return THZ_OK;
    StageTimer t(ctx, THZ_STAGE_INTENSITY);
    launch_intensity(ctx->stream, npix, ctx->plan_d.nt, const_cast<float *>(d_data), d_img, 0);
    return check_launch(ctx);
}

int thz_subtract_bias(thz_ctx *ctx, size_t npix, float *d_data, float *d_img)
{
    if (int rc = need_plan(ctx)) return rc;
    if (!d_data) return fail(ctx, THZ_ERR_INVALID, "thz_subtract_bias: null pointer");
    if (npix == 0) return THZ_OK;
    StageTimer t(ctx, THZ_STAGE_INTENSITY);
    launch_intensity(ctx->stream, npix, ctx->plan_d.nt, d_data, d_img, 1);
    return check_launch(ctx);
}

int thz_pixel_mean(thz_ctx *ctx, size_t nx, size_t ny, size_t len, int ncomp, const float *d_arr,
                   float *d_out)
{
    if (!ctx) return THZ_ERR_INVALID;
    if (int rc = use_device(ctx)) return rc;
    if (!d_arr || !d_out || nx == 0 || ny == 0 || len == 0 || (ncomp != 1 && ncomp != 2))
        return fail(ctx, THZ_ERR_INVALID, "thz_pixel_mean: bad argument");
    const size_t L = len * (size_t)ncomp;
    if (int rc = ensure_ws(ctx, ny * L * sizeof(float))) return rc;
    StageTimer t(ctx, THZ_STAGE_MEAN);
    float *acc = reinterpret_cast<float *>(ctx->ws);
    launch_sum_axis0(ctx->stream, d_arr, nx, ny * L, (float)nx, acc);
    launch_sum_axis0(ctx->stream, acc, ny, L, (float)ny, d_out);
    return check_launch(ctx);
}

int thz_pixel_sum(thz_ctx *ctx, size_t npix, size_t len, int ncomp, const float *d_arr,
                  float *d_out)
{
    if (!ctx) return THZ_ERR_INVALID;
    if (int rc = use_device(ctx)) return rc;
    if (!d_arr || !d_out || npix == 0 || len == 0 || (ncomp != 1 && ncomp != 2))
        return fail(ctx, THZ_ERR_INVALID, "thz_pixel_sum: bad argument");
    return pixel_sum_rows(ctx, d_arr, nullptr, npix, len * (size_t)ncomp, d_out);
}

}  // extern "C"

// Σ over the rows list[0 .. npix) of d_arr (list null: rows 0 .. npix - 1), rows of L floats; order-free (parallel)
// sums — the pixel sums behind the fast means and, with a list, a region of interest's (session_roi.cpp)
int pixel_sum_rows(thz_ctx *ctx, const float *d_arr, const uint32_t *d_list, size_t npix, size_t L, float *d_out)
{
    StageTimer t(ctx, d_list ? THZ_STAGE_ROI : THZ_STAGE_MEAN);
    if (npix < 64) {
        if (d_list) launch_gather_sum(ctx->stream, d_arr, L, d_list, (uint32_t)npix, 0.0f, d_out);
        else launch_sum_axis0(ctx->stream, d_arr, npix, L, 0.0f, d_out);
        return check_launch(ctx);
    }
    // two-level: row groups x column tiles with 4 rows of loads in flight per
    // thread, then one small pass over the partial rows
    const size_t max_groups = 2048, mid_groups = 32;
    if (int rc = ensure_ws(ctx, (max_groups + mid_groups) * L * sizeof(float))) return rc;
    float *part = reinterpret_cast<float *>(ctx->ws);
    float *part2 = part + max_groups * L;
    size_t groups = launch_colsum_partial(ctx->stream, d_arr, npix, L, part, max_groups, d_list);
    if (groups == 0) {  // very long rows: plain strided sum
        if (d_list) launch_gather_sum(ctx->stream, d_arr, L, d_list, (uint32_t)npix, 0.0f, d_out);
        else launch_sum_axis0(ctx->stream, d_arr, npix, L, 0.0f, d_out);
        return check_launch(ctx);
    }
    // the last level is one thread per column walking the partial rows one by one:
    // keep it short (a 2048-row walk is 2048 dependent loads)
    const float *src = part;
    if (groups > 4 * mid_groups) {
        groups = launch_colsum_partial(ctx->stream, part, groups, L, part2, mid_groups);
        src = part2;
    }
    launch_sum_axis0(ctx->stream, src, groups, L, 0.0f, d_out);
    return check_launch(ctx);
}

extern "C" {

int thz_roi_mask(thz_ctx *ctx, const uint64_t *poly_xy, size_t n_vertices, uint64_t scaling,
                 size_t shape0, size_t shape1, uint8_t *d_mask)
{
    if (!ctx) return THZ_ERR_INVALID;
    if (int rc = use_device(ctx)) return rc;
    if (!d_mask || shape0 == 0 || shape1 == 0 || scaling == 0 || (n_vertices && !poly_xy))
        return fail(ctx, THZ_ERR_INVALID, "thz_roi_mask: bad argument");
    // math_tools.rs:604-637: scale vertices (integer division), bounding box, clamp
    std::vector<uint64_t> poly(2 * (n_vertices ? n_vertices : 1), 0);
    uint64_t x_min = UINT64_MAX, y_min = UINT64_MAX, x_max = 0, y_max = 0;
    for (size_t i = 0; i < n_vertices; ++i) {
        const uint64_t x = poly_xy[2 * i] / scaling, y = poly_xy[2 * i + 1] / scaling;
        poly[2 * i] = x;
        poly[2 * i + 1] = y;
        if (x < x_min) x_min = x;
        if (y < y_min) y_min = y;
        if (x > x_max) x_max = x;
        if (y > y_max) y_max = y;
    }
    const uint64_t x_size = shape1, y_size = shape0;
    if (x_min > x_size - 1) x_min = x_size - 1;
    if (y_min > y_size - 1) y_min = y_size - 1;
    if (x_max > x_size - 1) x_max = x_size - 1;
    if (y_max > y_size - 1) y_max = y_size - 1;
    if (int rc = ensure_ws(ctx, poly.size() * sizeof(uint64_t))) return rc;
    StageTimer t(ctx, THZ_STAGE_ROI);
    HIP_TRY(ctx, hipMemcpyAsync(ctx->ws, poly.data(), poly.size() * sizeof(uint64_t),
                                hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // `poly` is a stack-lifetime vector
    launch_roi_mask(ctx->stream, reinterpret_cast<const uint64_t *>(ctx->ws), (int)n_vertices, x_min,
                    x_max, y_min, y_max, x_size, y_size, d_mask);
    if (int rc = check_launch(ctx)) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // workspace is reused by the next call
    return THZ_OK;
}

int thz_roi_mean(thz_ctx *ctx, const float *d_arr, size_t shape0, size_t shape1, size_t len,
                 const uint8_t *d_mask, float *d_out, uint32_t *d_count, int sum_only)
{
    if (!ctx) return THZ_ERR_INVALID;
    if (int rc = use_device(ctx)) return rc;
    if (!d_arr || !d_mask || !d_out || shape0 == 0 || shape1 == 0 || len == 0)
        return fail(ctx, THZ_ERR_INVALID, "thz_roi_mean: bad argument");
    std::vector<uint8_t> mask(shape0 * shape1);
    HIP_TRY(ctx, hipMemcpyAsync(mask.data(), d_mask, mask.size(), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // math_tools.rs:640-651: y outer, x inner; sampled pixel is [shape0 - y - 1, x]
    std::vector<uint32_t> list;
    for (size_t y = 0; y < shape0; ++y)
        for (size_t x = 0; x < shape1; ++x)
            if (mask[y * shape1 + x]) list.push_back((uint32_t)((shape0 - y - 1) * shape1 + x));
    const uint32_t count = (uint32_t)list.size();
    if (d_count) HIP_TRY(ctx, hipMemcpyAsync(d_count, &count, sizeof(count), hipMemcpyHostToDevice, ctx->stream));
    StageTimer t(ctx, THZ_STAGE_ROI);
    if (count == 0) {
        HIP_TRY(ctx, hipMemsetAsync(d_out, 0, len * sizeof(float), ctx->stream));  // :656-658
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return THZ_OK;
    }
    if (int rc = ensure_ws(ctx, list.size() * sizeof(uint32_t))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->ws, list.data(), list.size() * sizeof(uint32_t),
                                hipMemcpyHostToDevice, ctx->stream));
    launch_gather_sum(ctx->stream, d_arr, len, reinterpret_cast<const uint32_t *>(ctx->ws), count,
                      sum_only ? 0.0f : (float)count, d_out);
    if (int rc = check_launch(ctx)) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return THZ_OK;
}

int thz_scale3d(thz_ctx *ctx, const float *d_arr, size_t nx, size_t ny, size_t len, int ncomp,
                size_t s, float *d_out)
{
    if (!ctx) return THZ_ERR_INVALID;
    if (int rc = use_device(ctx)) return rc;
    if (!d_arr || !d_out || s == 0 || nx / s == 0 || ny / s == 0 || len == 0 ||
        (ncomp != 1 && ncomp != 2))
        return fail(ctx, THZ_ERR_INVALID, "thz_scale3d: bad argument");
    launch_scale3d(ctx->stream, d_arr, nx, ny, len * (size_t)ncomp, s, d_out);
    return check_launch(ctx);
}

int thz_tilt_apply(thz_ctx *ctx, size_t npix, const float *d_in, size_t nt_in, const float *d_taper,
                   const int32_t *d_insert_index, size_t nt_out, float *d_out)
{
    if (!ctx) return THZ_ERR_INVALID;
    if (int rc = use_device(ctx)) return rc;
    if (!d_in || !d_taper || !d_insert_index || !d_out || nt_in == 0 || nt_out < nt_in)
        return fail(ctx, THZ_ERR_INVALID, "thz_tilt_apply: bad argument");
    if (npix == 0) return THZ_OK;
    StageTimer t(ctx, THZ_STAGE_TD_WINDOW);
    launch_tilt(ctx->stream, npix, (int)nt_in, (int)nt_out, d_in, d_taper, d_insert_index, d_out);
    return check_launch(ctx);
}

int thz_synth_cube(thz_ctx *ctx, float *d_out, size_t ntraces, uint64_t first_trace,
                   const float *d_time, uint32_t seed, int subtract_bias)
{
    if (int rc = need_plan(ctx)) return rc;
    if (!d_out || !d_time) return fail(ctx, THZ_ERR_INVALID, "thz_synth_cube: null pointer");
    if (ntraces == 0) return THZ_OK;
    launch_synth(ctx->stream, d_out, ntraces, ctx->plan_d.nt, first_trace, d_time, seed,
                 subtract_bias);
    return check_launch(ctx);
}

int thz_enable_timing(thz_ctx *ctx, int enable)
{
    if (!ctx) return THZ_ERR_INVALID;
    if (enable < 0 || enable > 2) return THZ_ERR_INVALID;
    ctx->timing = enable;
    return THZ_OK;
}

int thz_timing_collect(thz_ctx *ctx, int stage, uint64_t *total_ns, uint64_t *count)
{
    if (!ctx || !total_ns || !count || stage < 0 || stage >= THZ_STAGE_COUNT) return THZ_ERR_INVALID;
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t tot = 0, n = 0;
    std::vector<thz_ctx::Rec> keep;
    for (auto &r : ctx->recs) {
        if (r.stage != stage) {
            keep.push_back(r);
            continue;
        }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            tot += (uint64_t)((double)ms * 1e6);
            ++n;
        }
        ctx->pool.push_back(r.a);
        ctx->pool.push_back(r.b);
    }
    ctx->recs.swap(keep);
    *total_ns = tot;
    *count = n;
    return THZ_OK;
}

int thz_stage_time_ns(thz_ctx *ctx, int stage, uint64_t *ns)
{
    if (!ctx || !ns || stage < 0 || stage >= THZ_STAGE_COUNT) return THZ_ERR_INVALID;
    *ns = ctx->stage_ns[stage];
    return THZ_OK;
}

}  // extern "C"
