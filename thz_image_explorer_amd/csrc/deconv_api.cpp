// deconv_api.cpp — C ABI of the frequency-dependent Richardson–Lucy
// deconvolution (K12): Deconvolution::filter, src/filters/deconvolution.rs:766-1041.
#include "ctx.hpp"
#include "deconv_host.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <functional>

using namespace thz;

namespace {

// The call's device blocks come from the context's pool (ctx.hpp: dc_pool) and go back to it when the call ends;
// everything the call enqueued has either been waited for by then or sits on the context's stream, in front of
// whatever the next call enqueues.
struct DevFree {
    thz_ctx *ctx;
    bool sweep;  // false: a phase that allocates (almost) nothing must not free what the other phases park in the pool
    explicit DevFree(thz_ctx *c, bool sweep_ = true) : ctx(c), sweep(sweep_)
    {
        if (sweep)
            for (auto &b : ctx->dc_pool) b.used_this_call = false;
    }
    ~DevFree()
    {
        auto &pool = ctx->dc_pool;
        if (!sweep) {
            for (auto &b : pool) b.in_use = false;
            return;
        }
        for (size_t i = 0; i < pool.size();) {
            pool[i].in_use = false;
            if (!pool[i].used_this_call) {  // left over from another geometry
                (void)hipFree(pool[i].p);
                pool.erase(pool.begin() + (long)i);
            } else ++i;
        }
    }
    template <class T>
    hipError_t alloc(T **out, size_t bytes)
    {
        if (bytes == 0) bytes = 16;
        thz_ctx::Block *best = nullptr;
        for (auto &b : ctx->dc_pool)  // the smallest free block that is large enough, and not wastefully so
            if (!b.in_use && b.bytes >= bytes && b.bytes <= 2 * bytes + 4096 && (!best || b.bytes < best->bytes)) best = &b;
        if (!best) {
            void *p = nullptr;
            hipError_t e = hipMalloc(&p, bytes);
            if (e != hipSuccess) {
                // out of memory with blocks of another geometry still parked in the pool (a call that freed everything at
                // its end would have succeeded here): give back every block nobody uses — and the chains' graphs, whose
                // nodes point into them — and try once more
                (void)hipGetLastError();
                (void)hipStreamSynchronize(ctx->stream);
                for (auto &g : ctx->dc_graph) g.drop();
                auto &pool = ctx->dc_pool;
                for (size_t i = 0; i < pool.size();) {
                    if (!pool[i].in_use) {
                        (void)hipFree(pool[i].p);
                        pool.erase(pool.begin() + (long)i);
                    } else ++i;
                }
                e = hipMalloc(&p, bytes);
            }
            if (e != hipSuccess) {
                *out = nullptr;
                return e;
            }
            ctx->dc_pool.push_back({p, bytes, false, false});
            best = &ctx->dc_pool.back();
        }
        best->in_use = best->used_this_call = true;
        *out = reinterpret_cast<T *>(best->p);
        return hipSuccess;
    }
};

int copy_through(thz_ctx *ctx, const float *d_in, float *d_out, float *d_img, size_t npix, size_t nt)
{
    if (d_out != d_in)
        HIP_TRY(ctx, hipMemcpyAsync(d_out, d_in, npix * nt * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    if (d_img) {
        launch_intensity(ctx->stream, npix, (int)nt, d_out, d_img, 0);
        if (int rc = check_launch(ctx)) return rc;
    }
    return THZ_OK;
}

}  // namespace

extern "C" {

int thz_host_psf_eval(const thz_psf *psf, const float *freqs, size_t n, float *wx, float *wy,
                      float *x0, float *y0)
{
    if (!psf || !freqs) return THZ_ERR_INVALID;
    for (size_t i = 0; i < n; ++i) {
        if (wx) wx[i] = hybrid_eval(psf->wx_fit, freqs[i]);
        if (wy) wy[i] = hybrid_eval(psf->wy_fit, freqs[i]);
        if (x0) x0[i] = spline_eval_const(psf->x0_spline, freqs[i]);
        if (y0) y0[i] = spline_eval_const(psf->y0_spline, freqs[i]);
    }
    return THZ_OK;
}

int thz_host_filter_bank(const float *time, size_t nt, const thz_deconv_cfg *cfg, float *filters,
                         float *centers)
{
    if (!time || nt < 2 || !cfg || !filters || !centers || cfg->n_filters < 2) return THZ_ERR_INVALID;
    std::vector<float> f, c;
    filter_bank((int)cfg->n_filters, (double)cfg->start_freq, (double)cfg->end_freq,
                (double)cfg->win_width, time, f, c);
    std::memcpy(filters, f.data(), f.size() * sizeof(float));
    std::memcpy(centers, c.data(), c.size() * sizeof(float));
    return THZ_OK;
}

int thz_host_band_psf(const thz_psf *psf, float center_freq, float dx, float dy, size_t img_rows,
                      size_t img_cols, float *out, size_t *rows, size_t *cols)
{
    if (!psf || !rows || !cols) return THZ_ERR_INVALID;
    const BandPsf b = band_psf(*psf, center_freq, dx, dy, (int)img_rows, (int)img_cols);
    *rows = (size_t)b.rows;
    *cols = (size_t)b.cols;
    if (out) std::memcpy(out, b.v.data(), b.v.size() * sizeof(float));
    return THZ_OK;
}

}  // extern "C"

namespace {

// The stage in ONE call (sp == nullptr: thz_deconvolve) or one of its three PHASES (ctx.hpp: thz_dc_*), which is
// how a group of GPUs runs it: the transform, the band energies and the recombination are per pixel, the iterations
// are per band over the whole image (SURVEY 8e's alternative; round 3).
//   phase 1  spectra + band energies of a slab's pixels, every band   -> sp->d_energy [n_filters][npix_local]
//   phase 2  Richardson-Lucy of the bands [band_begin, band_end) on their energy images over the whole grid
//            (sp->d_energy [bands][nx ny]) -> gains sp->d_gain [bands][nx ny]
//   phase 3  recombination of the slab's traces from every band's gains (sp->d_gain [n_filters][npix_local]) and the
//            spectra phase 1 left in the context -> d_out, d_img
// In the phases nothing is passed through: a guard makes phase 1 return THZ_SKIPPED, an abort makes phase 2 return
// THZ_ERR_ABORTED, and the caller hands every slab its own input.
struct DcSplit {
    int phase = 0;                 // 4: only the bands' costs (iterations x tiles), for dealing the bands out
    size_t npix_local = 0;
    float *d_energy = nullptr;
    float *d_gain = nullptr;
    std::vector<double> *costs = nullptr;
};

// THZ_DC_ENERGY_PV=0 (developer knob, A/B runs): band energies from one inverse transform per band (rounds 1-3)
// instead of the Parseval form
bool dc_pv_on()
{
    static const bool on = [] {
        const char *e = getenv("THZ_DC_ENERGY_PV");
        return !(e && e[0] == '0');
    }();
    return on;
}

// band energies of npix traces: the Parseval form where the context holds its tables (k_dc_energy_pv), else a
// transform per band
void dc_band_energies(thz_ctx *ctx, const PlanDev &P, size_t npix, int nt, int nbs, int shift, int nk, const float *d_in,
                      const c32 *d_spec, const c32 *d_H, float *d_energy)
{
    const thz_ctx::DcSpectra &SP = ctx->dc_spectra;
    if (SP.d_pv && dc_pv_on()) {
        const DcPvTables T{SP.d_pv, SP.d_pv + 512, SP.d_pv + 576, reinterpret_cast<const float *>(SP.d_pv + SP.pv_g_off),
                           SP.pv_gstride};
        launch_dc_energy_pv(ctx->stream, T, npix, nt, nbs, shift, nk, d_in, d_spec, d_energy);
        return;
    }
    launch_dc_energy(ctx->stream, P, npix, nt, nbs, shift, d_spec, d_H, d_energy);
}

// the recombination of npix traces: one launch where the padded length has an F core; elsewhere (M = 8192) the
// multiplier sum as a kernel of its own into a scratch of at most 256 MiB, chunk by chunk, and the generic transform
// behind it (THZ_DC_COMBINE_OLD, developer knob: everything in the generic kernel, as in rounds 1-3)
int dc_recombine(thz_ctx *ctx, DevFree &mem, const PlanDev &P, size_t npix, int nt, int nbs, int shift, size_t nk,
                 const c32 *d_spec, const c32 *d_H, const float *d_gain, float *d_out, float *d_img)
{
    if (dc_combine_has_f_core(P, nt, shift) || !dc_weight_spectra_supported(nbs) || getenv("THZ_DC_COMBINE_OLD")) {
        launch_dc_combine(ctx->stream, P, npix, nt, nbs, shift, d_spec, d_H, d_gain, d_out, d_img);
        return THZ_OK;
    }
    size_t chunk = ((size_t)256 << 20) / (nk * sizeof(c32));
    if (chunk < 1024) chunk = 1024;
    if (chunk > npix) chunk = npix;
    c32 *d_y = nullptr;
    HIP_TRY(ctx, mem.alloc(&d_y, chunk * nk * sizeof(c32)));
    for (size_t p0 = 0; p0 < npix; p0 += chunk) {
        const size_t n = std::min(chunk, npix - p0);
        launch_dc_weight_spectra(ctx->stream, n, npix, (int)nk, nbs, d_spec + p0 * nk, d_H, d_gain + p0, d_y);
        launch_dc_combine(ctx->stream, P, n, nt, 0, shift, d_y, nullptr, nullptr, d_out + p0 * (size_t)nt,
                          d_img ? d_img + p0 : nullptr);
    }
    return THZ_OK;
}

int deconvolve_impl(thz_ctx *ctx, const thz_psf *psf, const thz_deconv_cfg *cfg, size_t nx, size_t ny,
                    float dx, float dy, const float *d_in, float *d_out, float *d_img,
                    float *d_gains_out, volatile const int *abort_flag, float *progress, const DcSplit *sp)
{
    if (int rc = need_plan(ctx)) return rc;
    const int phase = sp ? sp->phase : 0;
    if (!psf || !cfg || nx == 0 || ny == 0 || ((phase == 0 || phase == 1) && !d_in) || ((phase == 0 || phase == 3) && !d_out))
        return fail(ctx, THZ_ERR_INVALID, "thz_deconvolve: bad argument");
    const size_t nt = ctx->time.size(), npix = nx * ny;
    const size_t npix_t = phase == 1 || phase == 3 ? sp->npix_local : npix;  // pixels the transforms of this call run over
    const int nb = (int)cfg->n_filters;
    if (progress) *progress = 0.0f;
    // developer knob: wall time of the call's phases on stderr (each tick drains the stream)
    const bool timing = getenv("THZ_DEBUG_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto tick = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "thz_deconvolve: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    // guards of the reference: each returns the input unchanged (:781-812, :873-885)
    bool skip = psf->wx_fit.correction.n_knots == 0 || nx < 16 || ny < 16 || nb < 2;
    std::vector<float> filters, centers;
    float w_min = 0.0f, w_max = 0.0f;
    if (!skip) {
        // the bank depends on the time axis and four numbers of the configuration: kept in the context
        thz_ctx::DcBank &K = ctx->dc_bank;
        if (K.n_filters != nb || K.start_freq != cfg->start_freq || K.end_freq != cfg->end_freq
            || K.win_width != cfg->win_width || K.time != ctx->time) {
            filter_bank(nb, (double)cfg->start_freq, (double)cfg->end_freq, (double)cfg->win_width,
                        ctx->time.data(), K.filters, K.centers);
            K.time = ctx->time;
            K.n_filters = nb; K.start_freq = cfg->start_freq; K.end_freq = cfg->end_freq; K.win_width = cfg->win_width;
            ++K.gen;
        }
        filters = K.filters;
        centers = K.centers;
        float wx_min = INFINITY, wx_max = -INFINITY, wy_min = INFINITY, wy_max = -INFINITY;
        for (int i = 0; i < nb; ++i) {
            const float wx = hybrid_eval(psf->wx_fit, centers[(size_t)i]);
            const float wy = hybrid_eval(psf->wy_fit, centers[(size_t)i]);
            wx_min = std::fmin(wx_min, wx); wx_max = std::fmax(wx_max, wx);
            wy_min = std::fmin(wy_min, wy); wy_max = std::fmax(wy_max, wy);
        }
        w_min = std::fmin(wx_min, wy_min);
        w_max = std::fmax(wx_max, wy_max);
        long mpx = (long)std::ceil(wx_max / dx) * 2 + 1, mpy = (long)std::ceil(wy_max / dy) * 2 + 1;
        if (mpx < 3) mpx = 3;
        if (mpy < 3) mpy = 3;
        if (mpx >= (long)ny || mpy >= (long)nx) skip = true;  // img_cols = ny, img_rows = nx
    }
    // Band-parallel calls (cfg->band_begin / band_end select bands [b0, b1) of the bank; 0, 0 = all): the outputs
    // of the ranks ADD UP to the full result.  That has to hold on the pass-through paths too — a guard, an
    // abort — so the input is copied through by the one rank that owns band 0 and every other rank contributes
    // zeros; a rank whose range is empty (more ranks than bands) contributes zeros as well.
    int b0 = (int)cfg->band_begin, b1 = (int)cfg->band_end;
    if ((b0 == 0 && b1 == 0) || phase == 1 || phase == 3 || phase == 4) { b0 = 0; b1 = nb; }
    if (b0 < 0 || b0 > b1 || (!skip && b1 > nb))
        return fail(ctx, THZ_ERR_INVALID, "thz_deconvolve: bad band range");
    const bool owns_first = b0 == 0;
    auto pass_through = [&]() -> int {
        if (owns_first) return copy_through(ctx, d_in, d_out, d_img, npix, nt);
        HIP_TRY(ctx, hipMemsetAsync(d_out, 0, npix * nt * sizeof(float), ctx->stream));
        if (d_img) HIP_TRY(ctx, hipMemsetAsync(d_img, 0, npix * sizeof(float), ctx->stream));
        if (d_gains_out && b1 > b0) HIP_TRY(ctx, hipMemsetAsync(d_gains_out, 0, (size_t)(b1 - b0) * npix * sizeof(float), ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return THZ_OK;
    };
    if (skip) {
        if (phase == 0)
            if (int rc = pass_through()) return rc;
        if (progress) *progress = 1.0f;
        return THZ_SKIPPED;
    }
    if (b0 == b1 && phase == 2) {  // a rank without a band of its own
        if (progress) *progress = 1.0f;
        return THZ_OK;
    }
    if (b0 == b1) {  // nothing to add from this rank
        HIP_TRY(ctx, hipMemsetAsync(d_out, 0, npix * nt * sizeof(float), ctx->stream));
        if (d_img) HIP_TRY(ctx, hipMemsetAsync(d_img, 0, npix * sizeof(float), ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (progress) *progress = 1.0f;
        return THZ_OK;
    }

    tick("filter bank, widths");
    // ---- transform plan for the padded length M
    size_t M = 1;
    while (M < nt + kDeconvTaps - 1) M <<= 1;
    // the tables of the padded length live in the context until M changes
    thz_ctx::DcPlan &PL = ctx->dc_plan;
    if (PL.M != M && phase != 2 && phase != 4) {
        if (PL.d_tw) (void)hipFree(PL.d_tw);
        PL = thz_ctx::DcPlan{};
        // (M = 16384: the generic recombination's three LDS buffers are 192 KiB — its launch was refused with
        // "invalid argument" — so traces of more than 7694 samples are refused here, by name)
        if (M > 8192 || !build_plan(M, PL.H, true))  // with the F core's tables where M has them (band energies)
            return fail(ctx, THZ_ERR_UNSUPPORTED, "thz_deconvolve: trace too long for the FIR transform");
        PlanHost &H = PL.H;
        std::vector<c32> pack(H.tw);
        pack.insert(pack.end(), H.tw_split.begin(), H.tw_split.end());
        PL.f = H.family == kFamilyF && !H.f_t1.empty();
        PL.o1 = pack.size();
        if (PL.f) pack.insert(pack.end(), H.f_t1.begin(), H.f_t1.end());
        PL.o2 = pack.size();
        if (PL.f) pack.insert(pack.end(), H.f_t2.begin(), H.f_t2.end());
        PL.o3 = pack.size();
        if (PL.f) pack.insert(pack.end(), H.f_w2n.begin(), H.f_w2n.end());
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&PL.d_tw), pack.size() * sizeof(c32)));
        HIP_TRY(ctx, hipMemcpyAsync(PL.d_tw, pack.data(), pack.size() * sizeof(c32), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // pack goes out of scope
        H.f_t1.clear();  // offsets only from here on
        H.f_t2.clear();
        H.f_w2n.clear();
        PL.M = M;
    }
    const PlanHost &H = PL.H;
    const size_t N = M / 2, nk = N + 1;
    DevFree mem(ctx, phase == 0 || phase == 2);
    const bool transforms = phase != 2 && phase != 4;
    c32 *d_tw = PL.d_tw, *d_spec = nullptr, *d_H = nullptr;
    float *d_energy = nullptr, *d_gain = nullptr, *d_ws = nullptr;
    RlBand *d_bands = nullptr;
    // the transform kernels of this call are the generic (LDS) ones; only the F core's tables ride along
    PlanDev P{};
    if (transforms) {  // (the iterations alone transform nothing)
        P = plan_dev(H, d_tw, d_tw + H.tw.size(), nullptr, nullptr);
        P.f_t1 = PL.f ? d_tw + PL.o1 : nullptr;
        P.f_t2 = PL.f ? d_tw + PL.o2 : nullptr;
        P.f_w2n = PL.f ? d_tw + PL.o3 : nullptr;
    }
    tick("plan, twiddles");
    // ---- filter spectra H_b[k] = (1/M) sum_j h_b[j] exp(-2 pi i j k / M), in double, for the bands of
    // this call (band-parallel multi-GPU: cfg->band_begin/band_end select a subset of the bank)
    const int nbs = b1 - b0;
    thz_ctx::DcSpectra &SP = ctx->dc_spectra;
    if (transforms && (!SP.d_H || SP.M != M || SP.bank_gen != ctx->dc_bank.gen || SP.b0 != b0 || SP.b1 != b1)) {
        if (SP.d_H) (void)hipFree(SP.d_H);
        if (SP.d_pv) (void)hipFree(SP.d_pv);
        SP = thz_ctx::DcSpectra{};
        std::vector<double> trig(2 * M);  // cos | sin of -2 pi m / M
        for (size_t m = 0; m < M; ++m) {
            const double a = -2.0 * 3.14159265358979323846 * (double)m / (double)M;
            trig[m] = std::cos(a);
            trig[M + m] = std::sin(a);
        }
        double *d_trig = nullptr;
        float *d_filters = nullptr;
        HIP_TRY(ctx, mem.alloc(&d_trig, trig.size() * sizeof(double)));
        HIP_TRY(ctx, mem.alloc(&d_filters, (size_t)nbs * kDeconvTaps * sizeof(float)));
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&SP.d_H), (size_t)nbs * nk * sizeof(c32)));
        HIP_TRY(ctx, hipMemcpyAsync(d_trig, trig.data(), trig.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_filters, filters.data() + (size_t)b0 * kDeconvTaps,
                                    (size_t)nbs * kDeconvTaps * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        launch_dc_filter_spectra(ctx->stream, d_filters, nbs, kDeconvTaps, d_trig, d_trig + M, (unsigned)M,
                                 (unsigned)nk, SP.d_H);
        // tables of the band energies' Parseval form: G_b = c_k M |H_b|^2 and the 512-point spectra of the two filter
        // halves that make the first / last `shift` samples of the full convolution (k_dc_energy_pv)
        std::vector<double> trig_e(2 * 512);
        std::vector<float> halves;
        if (dc_pv_on() && dc_energy_pv_supported(M, kDeconvTaps)) {
            const int s = (kDeconvTaps - 1) / 2;
            for (size_t m = 0; m < 512; ++m) {
                const double a = -2.0 * 3.14159265358979323846 * (double)m / 512.0;
                trig_e[m] = std::cos(a);
                trig_e[512 + m] = std::sin(a);
            }
            halves.resize((size_t)2 * nbs * s);
            for (int b = 0; b < nbs; ++b) {
                const float *h = filters.data() + (size_t)(b0 + b) * kDeconvTaps;
                std::copy(h, h + s, halves.begin() + (size_t)(2 * b) * s);
                std::copy(h + s + 1, h + 2 * s + 1, halves.begin() + (size_t)(2 * b + 1) * s);
            }
            std::vector<c32> t1, t2;
            dc_pv_core_tables(t1, t2);
            const int gstride = dc_pv_gstride((int)nk);
            const size_t n_tab = t1.size() + t2.size(), n_hpm = (size_t)nbs * 1024;
            double *d_trig_e = nullptr;
            float *d_halves = nullptr;
            c32 *d_hht = nullptr;
            HIP_TRY(ctx, mem.alloc(&d_trig_e, trig_e.size() * sizeof(double)));
            HIP_TRY(ctx, mem.alloc(&d_halves, halves.size() * sizeof(float)));
            HIP_TRY(ctx, mem.alloc(&d_hht, (size_t)2 * nbs * 512 * sizeof(c32)));
            HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&SP.d_pv),
                                   (n_tab + n_hpm) * sizeof(c32) + (size_t)nbs * gstride * sizeof(float)));
            SP.pv_g_off = n_tab + n_hpm;
            SP.pv_gstride = gstride;
            t1.insert(t1.end(), t2.begin(), t2.end());
            HIP_TRY(ctx, hipMemcpyAsync(SP.d_pv, t1.data(), n_tab * sizeof(c32), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(d_trig_e, trig_e.data(), trig_e.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(d_halves, halves.data(), halves.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
            launch_dc_filter_spectra(ctx->stream, d_halves, 2 * nbs, s, d_trig_e, d_trig_e + 512, 512u, 512u, d_hht);
            launch_dc_pv_tables(ctx->stream, nbs, (int)nk, gstride, M, SP.d_H, d_hht,
                                reinterpret_cast<float *>(SP.d_pv + SP.pv_g_off), SP.d_pv + n_tab);
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // t1 goes out of scope
        }
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // trig goes out of scope
        SP.M = M; SP.bank_gen = ctx->dc_bank.gen; SP.b0 = b0; SP.b1 = b1;
    }
    d_H = SP.d_H;
    centers = std::vector<float>(centers.begin() + b0, centers.begin() + b1);
    tick("filter spectra");
    const int shift = (kDeconvTaps - 1) / 2;
    if (phase == 1) {
        // ---- a slab's spectra (kept in the context for phase 3) and its band energies, every band
        thz_ctx::DcSlab &S = ctx->dc_slab;
        if (S.cap < npix_t * nk) {
            if (S.d_spec) {
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                (void)hipFree(S.d_spec);
            }
            S = thz_ctx::DcSlab{};
            HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&S.d_spec), npix_t * nk * sizeof(c32)));
            S.cap = npix_t * nk;
        }
        S.npix = npix_t; S.nk = nk; S.M = M;
        launch_dc_fft(ctx->stream, P, npix_t, (int)nt, d_in, S.d_spec);
        dc_band_energies(ctx, P, npix_t, (int)nt, nbs, shift, (int)nk, d_in, S.d_spec, d_H, sp->d_energy);
        if (int rc = check_launch(ctx)) return rc;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        tick("slab: transform, band energies");
        return THZ_OK;
    }
    if (phase == 3) {
        // ---- a slab's traces from every band's gains and the spectra phase 1 kept
        const thz_ctx::DcSlab &S = ctx->dc_slab;
        if (!S.d_spec || S.npix != npix_t || S.nk != nk || S.M != M)
            return fail(ctx, THZ_ERR_NOT_READY, "thz_dc_slab_combine: no spectra of this slab (thz_dc_slab_energies comes first)");
        if (int rc = dc_recombine(ctx, mem, P, npix_t, (int)nt, nbs, shift, nk, S.d_spec, d_H, sp->d_gain, d_out, d_img)) return rc;
        if (int rc = check_launch(ctx)) return rc;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        tick("slab: recombination");
        if (progress) *progress = 1.0f;
        return THZ_OK;
    }
    // ---- per-band PSFs, iteration counts, workspace layout
    std::vector<RlBand> bands((size_t)nbs);
    std::vector<float> psf_pack;
    size_t ws_floats = 0;
    unsigned blk = 0, tblk = 0;
    size_t tile_lds = 0;
    int max_iter = 0;
    std::vector<BandPsf> psfs((size_t)nbs);
    // the band PSFs are outer products of two profiles: their wide kernels run as two 1-D passes (k_rl_step_sep);
    // THZ_RL_NO_SEPARABLE (developer knob, for A/B timing) keeps the 2-D sums of k_rl_step_tiled<true>
    const bool separable = getenv("THZ_RL_NO_SEPARABLE") == nullptr;
    // ... and so do the narrow ones (<= 256 taps, the reference's direct sums) unless THZ_RL_NARROW_EXACT is set: the same
    // sums in another order (pr + pc multiply-adds per pixel instead of pr pc, tiles of 32 x 32 pixels), equal to the
    // reference's to rounding and inside the 1e-5 of the end-to-end tests; with the knob the reference's own order of
    // every sum, bit for bit (k_rl_step_tiled<false>)
    const bool narrow_sep = separable && getenv("THZ_RL_NARROW_EXACT") == nullptr;
    std::vector<char> sep_band((size_t)nbs, 0);
    for (int b = 0; b < nbs; ++b) {
        psfs[(size_t)b] = band_psf(*psf, centers[(size_t)b], dx, dy, (int)nx, (int)ny);
        const BandPsf &bp = psfs[(size_t)b];
        RlBand &B = bands[(size_t)b];
        B.h = (int)nx; B.w = (int)ny;
        B.pr = bp.rows; B.pc = bp.cols;
        B.pad_y = bp.rows / 2; B.pad_x = bp.cols / 2;
        B.H = B.h + 2 * B.pad_y; B.W = B.w + 2 * B.pad_x;
        B.mode = (bp.rows * bp.cols <= 256) ? 0 : 1;
        // n_iter, deconvolution.rs:969-971 (NaN -> 0 through `as usize`)
        const float fi = std::floor((bp.wx - w_min) / (w_max - w_min) * ((float)cfg->n_iterations - 1.0f) + 1.0f);
        B.n_iter = (fi != fi || fi < 0.0f) ? 0 : (int)fi;
        if (B.n_iter > max_iter) max_iter = B.n_iter;
        B.blk0 = blk;
        blk += (unsigned)(((size_t)B.H * B.W + 255) / 256);
        const bool sep_b = separable && (B.mode != 0 || (narrow_sep && (int)bp.fx.size() == bp.rows && (int)bp.fy.size() == bp.cols
                                                           && (bp.rows & 1) && (bp.cols & 1)));
        sep_band[(size_t)b] = sep_b;
        const int tile_kind = sep_b ? kRlSeparable : kRlNarrow;
        const int tile_rows = rl_tile_rows(tile_kind), tile_cols = rl_tile_cols(tile_kind);
        B.tiles_w = (B.W + tile_cols - 1) / tile_cols;
        B.n_tiles = B.tiles_w * ((B.H + tile_rows - 1) / tile_rows);
        tile_lds = std::max(tile_lds, rl_tile_lds_bytes(B.pr, B.pc, sep_b));
        const size_t img = (size_t)B.H * B.W;
        B.off_d = (unsigned)ws_floats; ws_floats += img;
        B.off_u = (unsigned)ws_floats; ws_floats += img;
        B.off_t = (unsigned)ws_floats; ws_floats += img;
    }
    if (phase == 4) {  // what the bands' iterations cost is a function of these: iterations | iterations x tiles, per band
        sp->costs->assign(2 * (size_t)nbs, 0.0);
        for (int b = 0; b < nbs; ++b) {
            const RlBand &B = bands[(size_t)b];
            (*sp->costs)[2 * (size_t)b] = (double)B.n_iter;
            (*sp->costs)[2 * (size_t)b + 1] = (double)B.n_iter * (double)rl_tile_block_count(B.pr, B.pc, (unsigned)B.n_tiles);
        }
        return THZ_OK;
    }
    // The bands never exchange anything, so the tiled iteration runs as independent CHAINS of launches, one stream
    // each: the wide-kernel bands (more than 256 taps) and the narrow-kernel ones (one dependent sum per pixel: a
    // launch of 17 us whatever its number of tiles, which under one common launch made every wide tile wait).
    // Inside a chain the bands are listed by falling iteration count, so the tiles that still iterate at iteration
    // `it` are a PREFIX of the list (live_blocks).  tblk0 counts from the start of the band's own list.
    // Measured on 128 x 128 x 1001, 500 iterations (profiles/r02_rl_chains.txt): one chain for everything 22.6 ms
    // per call, wide | narrow 16.6 ms, and the wide bands further split into three chains by iteration count
    // (THZ_RL_SPLIT_WIDE, kept as a developer knob) 19.3 ms — while every band is alive the chip is full (1 100
    // blocks for 512 places), the chains' launches take turns, and more chains only add launches.
    constexpr int kRlChains = 4;
    struct TileList {
        int kind = kRlNarrow;
        std::vector<int> order;                           // bands of this chain, by falling n_iter
        std::vector<std::pair<int, unsigned>> live_steps;  // (n_iter of a band, blocks up to and including it)
        unsigned blocks = 0, first = 0;                   // blocks of the list / where it starts in d_tiles
        size_t lds = 0;
        unsigned live_blocks(int it) const
        {
            unsigned n = 0;
            for (const auto &st : live_steps)
                if (st.first > it) n = st.second;
            return n;
        }
    };
    std::vector<TileList> lists;
    {
        std::vector<int> wide_bands, narrow_bands;
        // narrow bands that run as two 1-D passes keep a chain of their own (their tile list is long and short-lived: at
        // 512 x 512 pixels 18 bands x 304 tiles, most of which have left after 25 iterations)
        std::vector<int> narrow_sep_bands;
        for (int b = 0; b < nbs; ++b)
            (bands[(size_t)b].mode != 0 ? wide_bands : sep_band[(size_t)b] ? narrow_sep_bands : narrow_bands).push_back(b);
        auto by_iter = [&](int a, int c) { return bands[(size_t)a].n_iter > bands[(size_t)c].n_iter; };
        std::stable_sort(wide_bands.begin(), wide_bands.end(), by_iter);
        std::stable_sort(narrow_bands.begin(), narrow_bands.end(), by_iter);
        std::stable_sort(narrow_sep_bands.begin(), narrow_sep_bands.end(), by_iter);
        // The wide bands in three chains by iteration count ({b0}, {b1, b2}, the rest) when their tiles are many: every
        // chain's launches then reserve the LDS of ITS widest kernel only (the 47 x 57 band's 42 KB put three blocks on a CU
        // whatever band a block belongs to) and pick their block size by their own tile count.  Measured with the separable
        // kernels, call in ms, one chain / three: 128 x 128 pixels 7.2 / 8.4, 256 x 256 11.7 / 11.6, 384 x 384 18.9 / 17.6,
        // 512 x 512 27.5 / 25.8 (round 2's 2-D kernels at 128 x 128: 16.6 / 19.3) — so from 1024 wide tiles on.
        // THZ_RL_SPLIT_WIDE=0 / 1 (developer knob) forces one / three.
        size_t wide_tiles = 0;
        for (int b : wide_bands) wide_tiles += (size_t)bands[(size_t)b].n_tiles;
        const char *split_env = getenv("THZ_RL_SPLIT_WIDE");
        const bool split_wide = split_env ? split_env[0] != '0' : wide_tiles >= 1024;
        const size_t cut[3] = {split_wide ? 1 : wide_bands.size(), 3, wide_bands.size()};
        size_t at = 0;
        for (size_t end : cut) {
            end = std::min(end, wide_bands.size());
            if (end <= at) continue;
            TileList L;
            L.kind = separable ? kRlSeparable : kRlWide;
            L.order.assign(wide_bands.begin() + (long)at, wide_bands.begin() + (long)end);
            lists.push_back(L);
            at = end;
        }
        if (!narrow_sep_bands.empty()) {
            TileList L;
            L.kind = kRlSeparable;
            L.order = narrow_sep_bands;
            lists.push_back(L);
        }
        if (!narrow_bands.empty() && lists.size() < (size_t)kRlChains) {
            TileList L;
            L.order = narrow_bands;
            lists.push_back(L);
        } else if (!narrow_bands.empty()) {
            return fail(ctx, THZ_ERR_UNSUPPORTED, "thz_deconvolve: more chains than streams (THZ_RL_SPLIT_WIDE with mixed narrow bands)");
        }
    }
    static_assert(kRlChains >= 4, "three wide chains and the narrow one");
    for (TileList &L : lists) {
        L.first = tblk;
        for (int b : L.order) {
            RlBand &B = bands[(size_t)b];
            B.tblk0 = L.blocks;
            L.blocks += rl_tile_block_count(B.pr, B.pc, (unsigned)B.n_tiles);
            L.live_steps.emplace_back(B.n_iter, L.blocks);
            L.lds = std::max(L.lds, rl_tile_lds_bytes(B.pr, B.pc, L.kind == kRlSeparable));
        }
        tblk += L.blocks;
    }
    for (int b = 0; b < nbs; ++b) {
        const BandPsf &bp = psfs[(size_t)b];
        RlBand &B = bands[(size_t)b];
        B.off_psf = (unsigned)(ws_floats + psf_pack.size());
        psf_pack.insert(psf_pack.end(), bp.v.begin(), bp.v.end());
        B.off_mirror = (unsigned)(ws_floats + psf_pack.size());
        for (size_t i = bp.v.size(); i-- > 0;) psf_pack.push_back(bp.v[i]);  // psf[::-1, ::-1]
        B.off_fx = (unsigned)(ws_floats + psf_pack.size());
        psf_pack.insert(psf_pack.end(), bp.fx.begin(), bp.fx.end());
        B.off_fy = (unsigned)(ws_floats + psf_pack.size());
        psf_pack.insert(psf_pack.end(), bp.fy.begin(), bp.fy.end());
    }
    if (getenv("THZ_DEBUG_BANDS"))  // developer knob: the band table on stderr
        for (int b = 0; b < nbs; ++b)
            fprintf(stderr, "band %2d  f=%.3f THz  psf %3d x %3d  n_iter %4d  tiles %u\n", b, centers[(size_t)b],
                    bands[(size_t)b].pr, bands[(size_t)b].pc, bands[(size_t)b].n_iter,
                    rl_tile_block_count(bands[(size_t)b].pr, bands[(size_t)b].pc, (unsigned)bands[(size_t)b].n_tiles));
    for (RlBand &B : bands) B.off_zero = (unsigned)(ws_floats + psf_pack.size());  // the first of the zeros below
    psf_pack.insert(psf_pack.end(), 32, 0.0f);  // the tiled step reads taps a whole chunk at a time
    HIP_TRY(ctx, mem.alloc(&d_ws, (ws_floats + psf_pack.size()) * sizeof(float)));
    HIP_TRY(ctx, hipMemcpyAsync(d_ws + ws_floats, psf_pack.data(), psf_pack.size() * sizeof(float),
                                hipMemcpyHostToDevice, ctx->stream));
    std::vector<RlTileRef> tiles(tblk);
    for (const TileList &L : lists)
        for (int b : L.order) {
            const RlBand &B = bands[(size_t)b];
            const unsigned n = rl_tile_block_count(B.pr, B.pc, (unsigned)B.n_tiles);
            for (unsigned t = 0; t < n; ++t) tiles[L.first + B.tblk0 + t] = RlTileRef{B};
        }
    RlTileRef *d_tiles = nullptr;
    HIP_TRY(ctx, mem.alloc(&d_tiles, tiles.size() * sizeof(RlTileRef)));
    HIP_TRY(ctx, hipMemcpyAsync(d_tiles, tiles.data(), tiles.size() * sizeof(RlTileRef), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, mem.alloc(&d_bands, bands.size() * sizeof(RlBand)));
    HIP_TRY(ctx, hipMemcpyAsync(d_bands, bands.data(), bands.size() * sizeof(RlBand), hipMemcpyHostToDevice, ctx->stream));
    if (phase == 2) {  // the energy images come from the group, the gains go back to it
        d_energy = sp->d_energy;
        d_gain = sp->d_gain;
    } else {
        HIP_TRY(ctx, mem.alloc(&d_spec, npix * nk * sizeof(c32)));
        HIP_TRY(ctx, mem.alloc(&d_energy, (size_t)nbs * npix * sizeof(float)));
        HIP_TRY(ctx, mem.alloc(&d_gain, (size_t)nbs * npix * sizeof(float)));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // host staging vectors go out of scope below

    tick("band PSFs, workspace");
    if (phase == 0) {
        launch_dc_fft(ctx->stream, P, npix, (int)nt, d_in, d_spec);
        dc_band_energies(ctx, P, npix, (int)nt, nbs, shift, (int)nk, d_in, d_spec, d_H, d_energy);
    }
    launch_rl_init(ctx->stream, d_bands, nbs, blk, npix, d_energy, d_ws);
    if (int rc = check_launch(ctx)) return rc;
    tick("transform, band energies");
    // Richardson-Lucy iterations: two dependent launches each, ~10 us of work per launch — the loop
    // is launch-bound.  A batch of kRlBatch iterations is captured once into a hipGraph and replayed;
    // the batch's first iteration number lives in device memory (d_it) so that one graph serves all
    // batches.  Bands that have finished return at once (iteration >= n_iter), which also covers
    // the tail of the last batch.  The abort flag is polled between batches.
    constexpr int kRlBatch = 32;
    // image tile + halo + taps fit in LDS (THZ_NO_TILE: developer knob, forces the fallback for tests)
    const bool tiled = tile_lds <= (size_t)150 * 1024 && !getenv("THZ_NO_TILE");
    if (tiled) {
        size_t lds_kind[3] = {0, 0, 0};
        for (const TileList &L : lists) lds_kind[L.kind] = std::max(lds_kind[L.kind], L.lds);
        for (int k = 0; k < 3; ++k)
            if (lds_kind[k]) prepare_rl_step_tiled(k, lds_kind[k]);
    }
    const bool use_graph = !getenv("THZ_NO_GRAPH");      // developer knobs: plain launches / the whole grid
    const bool compact = !getenv("THZ_RL_FULL_GRID");    // every time (serial path), for A/B timing
    // Iterations are launch-sized work (10-20 us a launch), so a batch of kRlBatch iterations of a chain is
    // captured once into a hipGraph — a plain sequence of 2 kRlBatch launches over the chain's whole tile list (a
    // finished band's block leaves after one scalar load; cutting the grid to the live prefix measured no gain,
    // profiles/r02_deconv_timing_compact_vs_full_grid.txt) — and replayed on the chain's own stream; the batch's
    // first iteration number lives in device memory (d_it[chain]) so that one graph serves all of a chain's batches.
    // One multi-branch graph for all chains was tried first: its instantiation cost 9 ms per distinct set of grids.
    // The chains meet at the end of every batch, where the abort flag is polled.
    const bool parallel = tiled && use_graph && max_iter > kRlBatch;
    hipStream_t chain_stream[kRlChains] = {ctx->stream, ctx->stream, ctx->stream, ctx->stream};
    if (parallel)
        for (size_t c = 1; c < lists.size(); ++c) {
            // (the other chains at the lowest stream priority, so that their blocks would only fill what the first chain
            // leaves idle: measured, no difference — 35.8 against 36.0 ms at 512 x 512 x 1001)
            if (!ctx->aux_streams[c - 1]) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->aux_streams[c - 1], hipStreamNonBlocking));
            chain_stream[c] = ctx->aux_streams[c - 1];
        }
    auto sync_chains = [&]() -> hipError_t {
        hipError_t first = hipSuccess;
        for (size_t c = 0; c < (parallel ? lists.size() : (size_t)1); ++c) {
            const hipError_t e = hipStreamSynchronize(chain_stream[c]);
            if (first == hipSuccess) first = e;
        }
        return first;
    };
    // every way out of this function — error returns included — waits for the chains first: `mem` (declared above,
    // destroyed after this guard) hands the blocks they work on back to the pool
    struct ChainsDrained {
        std::function<hipError_t()> wait;
        ~ChainsDrained() { (void)wait(); }
    } chains_drained{sync_chains};
    int *d_it = nullptr;
    HIP_TRY(ctx, mem.alloc(&d_it, kRlChains * sizeof(int)));
    struct ChainGraphs {
        bool tried[kRlChains] = {false, false, false, false};
        hipEvent_t start = nullptr;
        ~ChainGraphs()
        {
            if (start) (void)hipEventDestroy(start);
        }
    } cg;
    auto chain_launches = [&](size_t c, hipStream_t st, const int *it_base, int o0, int o1, unsigned grid) {
        const TileList &L = lists[c];
        for (int o = o0; o < o1; ++o)
            for (int step = 0; step < 2; ++step)
                launch_rl_step_tiled(st, L.kind, d_tiles + L.first, grid, L.lds, it_base, o, step, d_ws);
    };
    // The graph of a chain's batch lives in the context (ctx.hpp: dc_graph) and is reused by the next call when
    // everything its nodes hold is the same: kernel kind, grid, LDS size and the three pointers.
    auto graph_of = [&](size_t c) -> hipGraphExec_t {
        thz_ctx::ChainGraph &G = ctx->dc_graph[c];
        if (cg.tried[c]) return G.exec;
        cg.tried[c] = true;
        const TileList &L = lists[c];
        if (G.exec && G.kind == L.kind && G.blocks == L.blocks && G.lds == L.lds && G.tiles == d_tiles + L.first
            && G.it == d_it + c && G.ws == d_ws)
            return G.exec;
        G.drop();
        if (hipStreamBeginCapture(chain_stream[c], hipStreamCaptureModeThreadLocal) == hipSuccess) {
            chain_launches(c, chain_stream[c], d_it + c, 0, kRlBatch, L.blocks);
            if (hipStreamEndCapture(chain_stream[c], &G.graph) != hipSuccess || !G.graph
                || hipGraphInstantiate(&G.exec, G.graph, nullptr, nullptr, 0) != hipSuccess) {
                G.drop();
                (void)hipGetLastError();
            } else {
                G.kind = L.kind; G.blocks = L.blocks; G.lds = L.lds;
                G.tiles = d_tiles + L.first; G.it = d_it + c; G.ws = d_ws;
            }
        }
        return G.exec;
    };
    if (parallel) {  // the chains start behind the padded images
        HIP_TRY(ctx, hipEventCreateWithFlags(&cg.start, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventRecord(cg.start, ctx->stream));
        for (size_t c = 1; c < lists.size(); ++c) HIP_TRY(ctx, hipStreamWaitEvent(chain_stream[c], cg.start, 0));
    }
    auto aborted = [&]() -> int {
        HIP_TRY(ctx, sync_chains());
        if (phase == 0)
            if (int rc = pass_through()) return rc;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return fail(ctx, THZ_ERR_ABORTED, "thz_deconvolve: aborted");
    };
    if (parallel) {
        // batch k + 1 is enqueued before the host waits for batch k: the chains never run dry while the host
        // walks a graph's nodes.  The abort flag is polled once per batch (cancellable_loops semantics; a click
        // takes effect within two batches, ~1.5 ms).
        struct DoneEvents {
            hipEvent_t ev[kRlChains][2] = {};
            bool pending[kRlChains][2] = {};
            ~DoneEvents()
            {
                for (auto &row : ev)
                    for (hipEvent_t e : row)
                        if (e) (void)hipEventDestroy(e);
            }
        } done;
        for (size_t c = 0; c < lists.size(); ++c)
            for (int k = 0; k < 2; ++k) HIP_TRY(ctx, hipEventCreateWithFlags(&done.ev[c][k], hipEventDisableTiming));
        std::vector<hipEvent_t> marks[kRlChains];
        hipEvent_t t_start = nullptr;
        // The first chain (the widest kernels, most iterations) is the call's critical path and ends latency-bound — one
        // band left, about a tile per CU — while the other chains are throughput work that competes with it when all
        // start together.  THZ_RL_DELAY (developer knob, batches; default 0) starts them late, beside the first chain's
        // thin end instead of its full beginning: measured at 512 x 512 x 1001 for 0 / 2 / 4 / 6 / 8 / 10 batches, the
        // iterations take 23.9 / 24.0 / 23.9 / 24.0 / 23.9 / 23.7 ms — the narrow launches cost the same 5.5-5.9 ms
        // wherever they run (profiles/r03_rl_chain_placement.txt): the two chains' launches take turns, they do not overlap.
        int chain_batches[kRlChains] = {0, 0, 0, 0}, delay[kRlChains] = {0, 0, 0, 0};
        for (size_t c = 0; c < lists.size(); ++c)
            while (chain_batches[c] * kRlBatch < max_iter && lists[c].live_blocks(chain_batches[c] * kRlBatch) > 0) ++chain_batches[c];
        for (size_t c = 1; c < lists.size(); ++c) {
            const char *e = getenv("THZ_RL_DELAY");
            const int slack = chain_batches[0] - chain_batches[c];
            delay[c] = e ? atoi(e) : 0;
            if (delay[c] > slack) delay[c] = slack;
            if (delay[c] < 0) delay[c] = 0;
        }
        auto submit = [&](int batch) -> int {
            for (size_t c = 0; c < lists.size(); ++c) {
                if (batch < delay[c]) continue;
                const int base = (batch - delay[c]) * kRlBatch, end = std::min(base + kRlBatch, max_iter);
                if (base >= max_iter || lists[c].live_blocks(base) == 0) continue;
                // Chains other than the first are throughput work off the critical path: plain launches, each over
                // exactly the tiles that still iterate (a prefix of the list) — a replayed graph dispatches the whole
                // list every time, and at 512 x 512 pixels the narrow list is 19 602 blocks of which most have left
                // after the first few iterations (THZ_RL_EXACT=0, developer knob: the graph for every chain)
                static const bool exact = [] { const char *e = getenv("THZ_RL_EXACT"); return !(e && e[0] == '0'); }();
                if (exact && c > 0 && lists[c].blocks >= 4096) {  // (smaller lists: equal within the noise; 512 x 512: 24.1 -> 23.1 ms)
                    for (int it = base; it < end; ++it)
                        if (const unsigned live = lists[c].live_blocks(it)) chain_launches(c, chain_stream[c], nullptr, it, it + 1, live);
                } else if (hipGraphExec_t exec = graph_of(c)) {
                    HIP_TRY(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_it + c), base, 1, chain_stream[c]));
                    HIP_TRY(ctx, hipGraphLaunch(exec, chain_stream[c]));
                } else {
                    chain_launches(c, chain_stream[c], nullptr, base, end, lists[c].blocks);
                }
                HIP_TRY(ctx, hipEventRecord(done.ev[c][batch & 1], chain_stream[c]));
                done.pending[c][batch & 1] = true;
                if (timing) {  // developer knob: when every chain's batches ended, relative to the first submit
                    hipEvent_t e = nullptr;
                    HIP_TRY(ctx, hipEventCreate(&e));
                    HIP_TRY(ctx, hipEventRecord(e, chain_stream[c]));
                    marks[c].push_back(e);
                }
            }
            return THZ_OK;
        };
        if (timing) {
            HIP_TRY(ctx, hipEventCreate(&t_start));
            HIP_TRY(ctx, hipEventRecord(t_start, ctx->stream));
        }
        const int n_batches = (max_iter + kRlBatch - 1) / kRlBatch;
        if (abort_flag && *abort_flag) return aborted();
        if (int rc = submit(0)) return rc;
        for (int k = 0; k < n_batches; ++k) {
            if (k + 1 < n_batches)
                if (int rc = submit(k + 1)) return rc;
            for (size_t c = 0; c < lists.size(); ++c)
                if (done.pending[c][k & 1]) {
                    HIP_TRY(ctx, hipEventSynchronize(done.ev[c][k & 1]));
                    done.pending[c][k & 1] = false;
                }
            if (progress) *progress = (float)std::min((k + 1) * kRlBatch, max_iter) / (float)max_iter;
            if (abort_flag && *abort_flag && k + 1 < n_batches) return aborted();
        }
        HIP_TRY(ctx, sync_chains());
        if (int rc = check_launch(ctx)) return rc;
        if (timing) {
            for (size_t c = 0; c < lists.size(); ++c) {
                fprintf(stderr, "thz_deconvolve: chain %zu (%s, %u blocks, %zu bands) batches end at [ms]:", c,
                        lists[c].kind == kRlSeparable ? "separable" : lists[c].kind == kRlWide ? "wide" : "narrow", lists[c].blocks, lists[c].order.size());
                for (hipEvent_t e : marks[c]) {
                    float ms = 0.0f;
                    (void)hipEventElapsedTime(&ms, t_start, e);
                    fprintf(stderr, " %.2f", ms);
                    (void)hipEventDestroy(e);
                }
                fprintf(stderr, "\n");
            }
            (void)hipEventDestroy(t_start);
        }
    } else {
        for (int base = 0; base < max_iter; base += kRlBatch) {
            if (abort_flag && *abort_flag) return aborted();  // polled between batches
            const int end = std::min(base + kRlBatch, max_iter);
            for (int it = base; it < end; ++it) {
                if (!tiled) {
                    launch_rl_step(ctx->stream, d_bands, nbs, blk, nullptr, it, 0, d_ws);
                    launch_rl_step(ctx->stream, d_bands, nbs, blk, nullptr, it, 1, d_ws);
                } else {
                    for (size_t c = 0; c < lists.size(); ++c)
                        chain_launches(c, ctx->stream, nullptr, it, it + 1, compact ? lists[c].live_blocks(it) : lists[c].blocks);
                }
            }
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // keeps the abort poll honest
            if (int rc = check_launch(ctx)) return rc;
            if (progress) *progress = (float)end / (float)max_iter;
        }
    }
    tick("iterations");
    launch_dc_gain(ctx->stream, d_bands, nbs, npix, d_energy, d_ws, d_gain);
    if (phase == 0)
        if (int rc = dc_recombine(ctx, mem, P, npix, (int)nt, nbs, shift, nk, d_spec, d_H, d_gain, d_out, d_img)) return rc;
    if (int rc = check_launch(ctx)) return rc;
    if (d_gains_out)
        HIP_TRY(ctx, hipMemcpyAsync(d_gains_out, d_gain, (size_t)nbs * npix * sizeof(float),
                                    hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // temporaries are freed on return
    tick("gains, recombination");
    if (progress) *progress = 1.0f;
    return THZ_OK;
}

}  // namespace

int thz_dc_slab_energies(thz_ctx *ctx, const thz_psf *psf, const thz_deconv_cfg *cfg, size_t nx, size_t ny, float dx, float dy,
                         const float *d_in, size_t npix_local, float *d_energy)
{
    if (!ctx || !d_energy || npix_local == 0) return THZ_ERR_INVALID;
    DcSplit sp;
    sp.phase = 1; sp.npix_local = npix_local; sp.d_energy = d_energy;
    return deconvolve_impl(ctx, psf, cfg, nx, ny, dx, dy, d_in, nullptr, nullptr, nullptr, nullptr, nullptr, &sp);
}

int thz_dc_band_gains(thz_ctx *ctx, const thz_psf *psf, const thz_deconv_cfg *cfg, size_t nx, size_t ny, float dx, float dy,
                      float *d_energy, float *d_gain, volatile const int *abort_flag, float *progress)
{
    if (!ctx || !cfg) return THZ_ERR_INVALID;
    if (cfg->band_end > cfg->band_begin && (!d_energy || !d_gain)) return THZ_ERR_INVALID;
    DcSplit sp;
    sp.phase = 2; sp.d_energy = d_energy; sp.d_gain = d_gain;
    return deconvolve_impl(ctx, psf, cfg, nx, ny, dx, dy, nullptr, nullptr, nullptr, nullptr, abort_flag, progress, &sp);
}

int thz_dc_slab_combine(thz_ctx *ctx, const thz_psf *psf, const thz_deconv_cfg *cfg, size_t nx, size_t ny, float dx, float dy,
                        size_t npix_local, float *d_gain, float *d_out, float *d_img)
{
    if (!ctx || !d_gain || npix_local == 0) return THZ_ERR_INVALID;
    DcSplit sp;
    sp.phase = 3; sp.npix_local = npix_local; sp.d_gain = d_gain;
    return deconvolve_impl(ctx, psf, cfg, nx, ny, dx, dy, nullptr, d_out, d_img, nullptr, nullptr, nullptr, &sp);
}

int thz_dc_band_costs(thz_ctx *ctx, const thz_psf *psf, const thz_deconv_cfg *cfg, size_t nx, size_t ny, float dx, float dy,
                      std::vector<double> *costs)
{
    if (!ctx || !costs) return THZ_ERR_INVALID;
    DcSplit sp;
    sp.phase = 4; sp.costs = costs;
    return deconvolve_impl(ctx, psf, cfg, nx, ny, dx, dy, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &sp);
}

extern "C" {

int thz_deconvolve(thz_ctx *ctx, const thz_psf *psf, const thz_deconv_cfg *cfg, size_t nx, size_t ny,
                   float dx, float dy, const float *d_in, float *d_out, float *d_img,
                   float *d_gains_out, volatile const int *abort_flag, float *progress)
{
    return deconvolve_impl(ctx, psf, cfg, nx, ny, dx, dy, d_in, d_out, d_img, d_gains_out, abort_flag, progress, nullptr);
}

}  // extern "C"
