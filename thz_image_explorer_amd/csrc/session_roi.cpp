// session_roi.cpp — regions of interest and avg_in_fourier_space on the resident-cube session: what the
// reference's ifft stage computes per region with every recompute (math_tools.rs:473-543) and what its plot
// copy-out reads afterwards (data_thread.rs:1442-1482), taken from the resident arrays behind the fused launch.
//
//   mask     the reference's integer ray cast on usize coordinates (math_tools.rs:574-591, 604-637) — k_roi_mask,
//            bit for bit — once per set of regions and grid; kept as a list of the session's pixels inside, in
//            the reference's visiting order (y outer, x inner; mask position (x, y) samples pixel
//            [shape0 - y - 1, x], :640-651)
//   sums     per region: amplitudes and phases (nf), the chain's final traces and the ifft stage's input traces
//            (nt).  want_means == 2: one thread per sample walks the list in order (sequential f32 adds, the
//            reference's own; the means are then those of the resident arrays bit for bit).  Otherwise the
//            list's rows are added in parallel like the pixel sums.
//   group    a slab session lists the pixels of ITS rows of the whole grid's mask (the flipped row index runs
//            along the sharded axis); the slabs' sums are all-reduced by group_api.cpp between the two halves
//            below and divided by the whole grid's count.
#include "session.hpp"

#include <cmath>
#include <cstring>

using namespace thz;

namespace {

// block layout of d_roi_sum / d_roi for R regions: [amplitudes R x nf | phases R x nf | stage input R x nt | final R x nt]
struct RoiLayout {
    size_t R, nf, nt;
    size_t amp(size_t r) const { return r * nf; }
    size_t ph(size_t r) const { return R * nf + r * nf; }
    size_t src(size_t r) const { return 2 * R * nf + r * nt; }
    size_t fin(size_t r) const { return 2 * R * nf + R * nt + r * nt; }
    size_t total() const { return 2 * R * (nf + nt); }
};

RoiLayout layout(const thz_session *s) { return RoiLayout{s->rois.size(), s->nf_out, s->nt_out}; }

float *final_data(const thz_session *s) { return s->deconv_current ? s->d_deconv : s->d_data; }

// pixel lists of every region for the current grid (rebuilt only when the regions or the grid changed)
int build_lists(thz_session *s)
{
    thz_ctx *ctx = s->ctx;
    const size_t rows = s->nx_cur, cols = s->ny_cur;
    const size_t grid_rows = s->grid_rows ? s->grid_rows : rows, x0 = s->grid_rows ? s->grid_x0 : 0;
    uint8_t *d_mask = nullptr;
    std::vector<uint8_t> mask;
    std::vector<uint32_t> list;
    int rc = THZ_OK;
    for (SessionRoi &r : s->rois) {
        if (r.for_rows == rows && r.for_cols == cols && r.for_scale == s->scale && r.for_x0 == x0 && r.for_grid_rows == grid_rows)
            continue;
        if (!d_mask) {
            HIP_TRY(ctx, hipMalloc((void **)&d_mask, grid_rows * cols));
            mask.resize(grid_rows * cols);
        }
        // the whole grid's mask: the bounding box is clamped to the whole array's bounds (math_tools.rs:632-637)
        rc = thz_roi_mask(ctx, r.poly.data(), r.poly.size() / 2, (uint64_t)s->scale, grid_rows, cols, d_mask);
        if (rc) break;
        rc = thz_memcpy_d2h(ctx, mask.data(), d_mask, mask.size());
        if (rc) break;
        list.clear();
        uint32_t total = 0;
        for (size_t y = 0; y < grid_rows; ++y)
            for (size_t x = 0; x < cols; ++x)
                if (mask[y * cols + x]) {
                    ++total;
                    const size_t row = grid_rows - y - 1;  // :647
                    if (row >= x0 && row < x0 + rows) list.push_back((uint32_t)((row - x0) * cols + x));
                }
        if (list.size() > r.list_cap) {
            if (r.d_list) (void)hipFree(r.d_list);
            r.d_list = nullptr;
            r.list_cap = 0;
            if (hipMalloc((void **)&r.d_list, list.size() * sizeof(uint32_t)) != hipSuccess) {
                rc = fail(ctx, THZ_ERR_HIP, "region of interest: allocation of the pixel list failed");
                break;
            }
            r.list_cap = list.size();
        }
        if (!list.empty()) {
            rc = thz_memcpy_h2d(ctx, r.d_list, list.data(), list.size() * sizeof(uint32_t));
            if (rc) break;
        }
        r.count = (uint32_t)list.size();
        r.total = total;
        ++s->src_gen;  // other pixels: the kept sums of the source traces are void
        r.for_rows = rows; r.for_cols = cols; r.for_scale = s->scale; r.for_x0 = x0; r.for_grid_rows = grid_rows;
    }
    if (d_mask) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(d_mask);
    }
    return rc;
}

template <class T>
int dev_realloc(thz_ctx *ctx, T **p, size_t n)
{
    if (*p) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(*p); *p = nullptr; }
    HIP_TRY(ctx, hipMalloc((void **)p, (n ? n : 1) * sizeof(T)));
    return THZ_OK;
}

}  // namespace

size_t session_roi_floats(const thz_session *s) { return layout(s).total(); }

void session_roi_free(thz_session *s)
{
    for (SessionRoi &r : s->rois)
        if (r.d_list) (void)hipFree(r.d_list);
    s->rois.clear();
    for (void *p : {(void *)s->d_roi, (void *)s->d_roi_sum, (void *)s->d_wsep})
        if (p) (void)hipFree(p);
    s->d_roi = s->d_roi_sum = s->d_wsep = nullptr;
    s->roi_floats = s->wsep_floats = 0;
    s->have_rois = false;
}

int session_roi_sums(thz_session *s, const thz_chain_cfg *cfg, bool *data_only_io)
{
    bool data_only = *data_only_io;
    thz_ctx *ctx = s->ctx;
    s->have_rois = false;
    if (s->rois.empty() || !s->have_outputs) return THZ_OK;
    if (int rc = use_device(ctx)) return rc;
    if (int rc = build_lists(s)) return rc;
    const RoiLayout L = layout(s);
    if (s->roi_floats != L.total()) {
        s->roi_floats = 0;
        if (int rc = dev_realloc(ctx, &s->d_roi, L.total())) return rc;
        if (int rc = dev_realloc(ctx, &s->d_roi_sum, L.total())) return rc;
        s->roi_floats = L.total();
        s->roi_src_gen = 0;
        data_only = false;  // nothing to keep
    }
    *data_only_io = data_only;
    const bool ordered = cfg->want_means == 2;
    const float *fin = final_data(s);
    // the source block: always with the ordered sums (the multipliers are inside them), else only when the source
    // traces, the grid or the regions changed since it was last taken
    const bool src_now = !data_only && (ordered || s->roi_src_gen != s->src_gen);
    s->roi_src_fresh = src_now;
    if (src_now) s->roi_src_gen = ordered ? 0 : s->src_gen;
    for (size_t r = 0; r < L.R; ++r) {
        const SessionRoi &roi = s->rois[r];
        float *o = s->d_roi_sum;
        if (roi.count == 0) {  // none of this session's pixels inside: its share of every sum is zero
            if (!data_only) {
                HIP_TRY(ctx, hipMemsetAsync(o + L.amp(r), 0, L.nf * sizeof(float), ctx->stream));
                HIP_TRY(ctx, hipMemsetAsync(o + L.ph(r), 0, L.nf * sizeof(float), ctx->stream));
                if (src_now) HIP_TRY(ctx, hipMemsetAsync(o + L.src(r), 0, L.nt * sizeof(float), ctx->stream));
            }
            HIP_TRY(ctx, hipMemsetAsync(o + L.fin(r), 0, L.nt * sizeof(float), ctx->stream));
            continue;
        }
        if (ordered) {
            StageTimer t(ctx, THZ_STAGE_ROI);
            if (!data_only) {
                launch_gather_sum(ctx->stream, s->d_amp, L.nf, roi.d_list, roi.count, 0.0f, o + L.amp(r));
                launch_gather_sum(ctx->stream, s->d_ph, L.nf, roi.d_list, roi.count, 0.0f, o + L.ph(r));
                // the ifft stage's input traces = the source traces times the multipliers in front of the transform,
                // one f32 multiply each, in the chain's order
                launch_gather_sum_w(ctx->stream, s->d_src, L.nt, roi.d_list, roi.count, 0.0f,
                                    s->wsep_on[0] ? s->d_wsep : nullptr, s->wsep_on[1] ? s->d_wsep + L.nt : nullptr,
                                    s->wsep_on[2] ? s->d_wsep + 2 * L.nt : nullptr, o + L.src(r));
            }
            launch_gather_sum(ctx->stream, fin, L.nt, roi.d_list, roi.count, 0.0f, o + L.fin(r));
            if (int rc = check_launch(ctx)) return rc;
        } else {
            if (!data_only) {
                if (int rc = pixel_sum_rows(ctx, s->d_amp, roi.d_list, roi.count, L.nf, o + L.amp(r))) return rc;
                if (int rc = pixel_sum_rows(ctx, s->d_ph, roi.d_list, roi.count, L.nf, o + L.ph(r))) return rc;
                if (src_now)
                    if (int rc = pixel_sum_rows(ctx, s->d_src, roi.d_list, roi.count, L.nt, o + L.src(r))) return rc;
            }
            if (int rc = pixel_sum_rows(ctx, fin, roi.d_list, roi.count, L.nt, o + L.fin(r))) return rc;
        }
    }
    return THZ_OK;
}

int session_roi_finish(thz_session *s, const thz_chain_cfg *cfg, bool data_only)
{
    thz_ctx *ctx = s->ctx;
    if (s->rois.empty() || !s->have_outputs || s->roi_floats == 0) return THZ_OK;
    if (int rc = use_device(ctx)) return rc;
    const RoiLayout L = layout(s);
    const bool ordered = cfg->want_means == 2;
    const float *d_pre = s->d_vec;  // the composed multiplier in front of the transform (session_enqueue)
    for (size_t r = 0; r < L.R; ++r) {
        const SessionRoi &roi = s->rois[r];
        const float *in = s->d_roi_sum;
        float *o = s->d_roi;
        if (roi.total == 0) {  // empty mask: zeros (math_tools.rs:656-658)
            HIP_TRY(ctx, hipMemsetAsync(o + L.amp(r), 0, L.nf * sizeof(float), ctx->stream));
            HIP_TRY(ctx, hipMemsetAsync(o + L.ph(r), 0, L.nf * sizeof(float), ctx->stream));
            HIP_TRY(ctx, hipMemsetAsync(o + L.src(r), 0, L.nt * sizeof(float), ctx->stream));
            HIP_TRY(ctx, hipMemsetAsync(o + L.fin(r), 0, L.nt * sizeof(float), ctx->stream));
            continue;
        }
        const float n = (float)roi.total;
        if (!data_only) {
            launch_div_vec(ctx->stream, in + L.amp(r), nullptr, n, L.nf, o + L.amp(r));
            launch_div_vec(ctx->stream, in + L.ph(r), nullptr, n, L.nf, o + L.ph(r));
            // parallel sums are those of the SOURCE traces: the multiplier is the same for every pixel
            launch_div_vec(ctx->stream, in + L.src(r), ordered ? nullptr : d_pre, n, L.nt, o + L.src(r));
        }
        launch_div_vec(ctx->stream, in + L.fin(r), nullptr, n, L.nt, o + L.fin(r));
        if (int rc = check_launch(ctx)) return rc;
    }
    if (cfg->avg_in_fourier_space && !data_only) {
        // roi_data = C2R(from_polar(roi_signal_fft, roi_phase_fft), Im X[0] := 0) / nt per region (math_tools.rs:496-529)
        s->roi_polar.assign(L.R * L.nt, 0.0f);
        s->roi_polar_ok.assign(L.R, 0);
        std::vector<float> amp(L.nf), ph(L.nf);
        if (ctx->time.size() != L.nt || std::memcmp(ctx->time.data(), s->time_out.data(), L.nt * sizeof(float)) != 0)
            if (int rc = thz_set_time_axis(ctx, s->time_out.data(), L.nt)) return rc;
        for (size_t r = 0; r < L.R; ++r) {
            if (int rc = thz_memcpy_d2h(ctx, amp.data(), s->d_roi + L.amp(r), L.nf * sizeof(float))) return rc;
            if (int rc = thz_memcpy_d2h(ctx, ph.data(), s->d_roi + L.ph(r), L.nf * sizeof(float))) return rc;
            // realfft refuses a spectrum whose last bin (even length) has an imaginary part; the first bin's is
            // cleared by the reference itself (:510-512).  The reference then falls back to the traces (:530-538).
            const bool refused = L.nt % 2 == 0 && amp[L.nf - 1] * std::sin(ph[L.nf - 1]) != 0.0f;
            if (refused) continue;
            if (int rc = thz_polar_ifft(ctx, amp.data(), ph.data(), 1, s->roi_polar.data() + r * L.nt)) return rc;
            s->roi_polar_ok[r] = 1;
        }
    } else if (!cfg->avg_in_fourier_space) {
        s->roi_polar.clear();
        s->roi_polar_ok.clear();
    }
    s->have_rois = true;
    return THZ_OK;
}

int session_avg_data(thz_session *s, const thz_chain_cfg *cfg)
{
    s->avg_data.clear();
    if (!cfg->avg_in_fourier_space || !s->have_means) return THZ_OK;
    thz_ctx *ctx = s->ctx;
    const size_t nt = s->nt_out, nf = s->nf_out;
    std::vector<float> amp(nf), ph(nf), out(nt);
    if (int rc = thz_memcpy_d2h(ctx, amp.data(), s->d_avg + 2 * nf, nf * sizeof(float))) return rc;
    if (int rc = thz_memcpy_d2h(ctx, ph.data(), s->d_avg + 3 * nf, nf * sizeof(float))) return rc;
    if (ctx->time.size() != nt || std::memcmp(ctx->time.data(), s->time_out.data(), nt * sizeof(float)) != 0)
        if (int rc = thz_set_time_axis(ctx, s->time_out.data(), nt)) return rc;
    // math_tools.rs:442-470.  The reference unwrap()s the C2R here: where realfft refuses the spectrum (an
    // imaginary part in the first bin or in the last one of an even length) it panics; the transform realfft
    // has carried out by then ignores those imaginary parts, and that is what this returns.
    if (int rc = thz_polar_ifft(ctx, amp.data(), ph.data(), 0, out.data())) return rc;
    s->avg_data = out;
    return THZ_OK;
}

extern "C" {

int thz_session_set_rois(thz_session *s, size_t n_rois, const size_t *n_vertices, const uint64_t *poly_xy)
{
    if (!s || (n_rois && (!n_vertices || !poly_xy))) return THZ_ERR_INVALID;
    thz_ctx *ctx = s->ctx;
    if (int rc = use_device(ctx)) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (SessionRoi &r : s->rois)
        if (r.d_list) (void)hipFree(r.d_list);
    s->rois.clear();
    s->have_rois = false;
    s->roi_floats = 0;  // the next recompute takes every sum afresh, whatever its start position
    size_t off = 0;
    for (size_t i = 0; i < n_rois; ++i) {
        SessionRoi r;
        r.poly.assign(poly_xy + 2 * off, poly_xy + 2 * (off + n_vertices[i]));
        off += n_vertices[i];
        s->rois.push_back(std::move(r));
    }
    return THZ_OK;
}

size_t thz_session_roi_count(const thz_session *s) { return s ? s->rois.size() : 0; }

int thz_session_roi(thz_session *s, size_t roi, const thz_roi_out *out)
{
    if (!s || !out) return THZ_ERR_INVALID;
    thz_ctx *ctx = s->ctx;
    if (roi >= s->rois.size()) return fail(ctx, THZ_ERR_INVALID, "thz_session_roi: no such region");
    if (!s->have_rois) return fail(ctx, THZ_ERR_NOT_READY, "thz_session_roi: no recompute has run since the regions were set");
    const RoiLayout L = layout(s);
    if (out->count) *out->count = s->rois[roi].total;
    if (out->signal_fft)
        if (int rc = thz_memcpy_d2h(ctx, out->signal_fft, s->d_roi + L.amp(roi), L.nf * sizeof(float))) return rc;
    if (out->phase_fft)
        if (int rc = thz_memcpy_d2h(ctx, out->phase_fft, s->d_roi + L.ph(roi), L.nf * sizeof(float))) return rc;
    const bool polar = s->have_last_cfg && s->last_cfg.avg_in_fourier_space;
    const bool polar_ok = polar && roi < s->roi_polar_ok.size() && s->roi_polar_ok[roi];
    auto stage_roi_data = [&](float *dst) -> int {
        if (polar_ok) {
            std::memcpy(dst, s->roi_polar.data() + roi * L.nt, L.nt * sizeof(float));
            return THZ_OK;
        }
        return thz_memcpy_d2h(ctx, dst, s->d_roi + L.src(roi), L.nt * sizeof(float));
    };
    if (out->roi_data)
        if (int rc = stage_roi_data(out->roi_data)) return rc;
    if (out->signal) {
        // data_thread.rs:1445-1451, then :1476-1482 overwrites it with roi_data when averaging in Fourier space
        if (polar) {
            if (int rc = stage_roi_data(out->signal)) return rc;
        } else if (int rc = thz_memcpy_d2h(ctx, out->signal, s->d_roi + L.fin(roi), L.nt * sizeof(float))) return rc;
    }
    return THZ_OK;
}

}  // extern "C"
