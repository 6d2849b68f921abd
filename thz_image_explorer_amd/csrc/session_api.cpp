// session_api.cpp — resident cube + whole-chain recompute on top of the stage
// entry points (include/thzgpu.h, "session" section).
#include "session.hpp"
#include "host_windows.hpp"

#include <cstring>

using namespace thz;

namespace {

// the chain's final trace cube / image: the Deconvolution stage's output while it is current
float *final_data(const thz_session *s) { return s->deconv_current ? s->d_deconv : s->d_data; }
float *final_img(const thz_session *s) { return s->deconv_current ? s->d_deconv_img : s->d_img; }

template <class T>
int dev_alloc(thz_ctx *ctx, T **p, size_t n)
{
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    HIP_TRY(ctx, hipMalloc((void **)p, (n ? n : 1) * sizeof(T)));
    return THZ_OK;
}

int alloc_outputs(thz_session *s, size_t nt_out)
{
    thz_ctx *ctx = s->ctx;
    const size_t npix = s->nx_cur * s->ny_cur, nf = nt_out / 2 + 1;
    if (nt_out == s->nt_out && npix == s->out_pix && s->d_fft) return THZ_OK;
    s->out_pix = 0;
    if (int rc = dev_alloc(ctx, &s->d_fft, npix * nf * 2)) return rc;
    if (int rc = dev_alloc(ctx, &s->d_amp, npix * nf)) return rc;
    if (int rc = dev_alloc(ctx, &s->d_ph, npix * nf)) return rc;
    if (int rc = dev_alloc(ctx, &s->d_data, npix * nt_out)) return rc;
    if (int rc = dev_alloc(ctx, &s->d_avg, 4 * nf)) return rc;
    s->vec_floats = 3 * nt_out + 3 * nf + 16;  // pre | post | mask | cmask (+ tilt scratch)
    if (int rc = dev_alloc(ctx, &s->d_vec, s->vec_floats)) return rc;
    if (s->h_vec) {
        (void)hipHostFree(s->h_vec);
        s->h_vec = nullptr;
    }
    HIP_TRY(ctx, hipHostMalloc((void **)&s->h_vec, s->vec_floats * sizeof(float), hipHostMallocDefault));
    s->nt_out = nt_out;
    s->nf_out = nf;
    s->out_pix = npix;
    return THZ_OK;
}

}  // namespace

extern "C" {

int thz_chain_cfg_default(const float *time, size_t nt, thz_chain_cfg *out)
{
    if (!time || nt < 2 || !out) return THZ_ERR_INVALID;
    std::memset(out, 0, sizeof(*out));
    out->tilt_active = 1;
    out->td_before_active = 1;
    out->td_before_low = (double)time[0];          // reset(), band_pass_td_before_fft.rs:66-72
    out->td_before_high = (double)time[nt - 1];
    out->td_before_width = 2.0;
    out->fft_window = thz_window_cfg{THZ_WIN_ADAPTED_BLACKMAN, 1.0f, 7.0f};
    out->fd_active = 1;
    out->fd_low = 0.2; out->fd_high = 5.0; out->fd_width = 0.1;
    out->td_after_active = 1;
    out->td_after_low = (double)time[0];
    out->td_after_high = (double)time[nt - 1];
    out->td_after_width = 0.1;
    out->want_means = 1;
    out->scale_factor = 1;  // ConfigContainer.scale_factor, config.rs:193-212
    return THZ_OK;
}

int thz_session_create(thz_ctx *ctx, size_t nx, size_t ny, size_t nt, const float *time, float dx,
                       float dy, thz_session **out)
{
    if (!ctx || !out || !time || nx == 0 || ny == 0 || nt < 2) return THZ_ERR_INVALID;
    *out = nullptr;
    if (int rc = thz_set_time_axis(ctx, time, nt)) return rc;
    thz_session *s = new thz_session();
    s->ctx = ctx; s->nx = nx; s->ny = ny; s->nt = nt; s->nf = nt / 2 + 1; s->dx = dx; s->dy = dy;
    s->nx_cur = nx; s->ny_cur = ny; s->dx_cur = dx; s->dy_cur = dy;
    s->time.assign(time, time + nt);
    s->time_out = s->time;
    int rc = dev_alloc(ctx, &s->d_raw, nx * ny * nt);
    if (!rc) rc = dev_alloc(ctx, &s->d_img, nx * ny);
    if (!rc) rc = dev_alloc(ctx, &s->d_rawsum, nt);
    if (!rc) rc = alloc_outputs(s, nt);
    if (rc) { thz_session_destroy(s); return rc; }
    *out = s;
    return THZ_OK;
}

void thz_session_destroy(thz_session *s)
{
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    for (void *p : {(void *)s->d_raw, (void *)s->d_fft, (void *)s->d_amp, (void *)s->d_ph, (void *)s->d_data,
                    (void *)s->d_img, (void *)s->d_avg, (void *)s->d_vec, (void *)s->d_tilt, (void *)s->d_ins,
                    (void *)s->d_opacity, (void *)s->d_deconv, (void *)s->d_deconv_img, (void *)s->d_scaled,
                    (void *)s->d_rawsum, (void *)s->d_msum, (void *)s->d_carry_in, (void *)s->d_carry_out})
        if (p) (void)hipFree(p);
    if (s->h_vec) (void)hipHostFree(s->h_vec);
    session_roi_free(s);
    delete s;
}

int thz_session_upload(thz_session *s, const float *cube, int subtract_bias)
{
    if (!s) return THZ_ERR_INVALID;
    thz_ctx *ctx = s->ctx;
    if (int rc = use_device(ctx)) return rc;
    const size_t npix = s->nx * s->ny;
    if (int rc = thz_set_time_axis(ctx, s->time.data(), s->nt)) return rc;
    // the image below is the raw grid's: outputs of an earlier (possibly scaled) recompute are void
    s->have_outputs = false; s->have_means = false; s->deconv_current = false;
    s->scale = 1; s->nx_cur = s->nx; s->ny_cur = s->ny; s->dx_cur = s->dx; s->dy_cur = s->dy;
    ++s->src_gen;  // new source traces: the regions' kept sums of them are void
    if (cube) HIP_TRY(ctx, hipMemcpyAsync(s->d_raw, cube, npix * s->nt * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    launch_intensity(ctx->stream, npix, (int)s->nt, s->d_raw, s->d_img, subtract_bias ? 1 : 0);
    if (int rc = check_launch(ctx)) return rc;
    // Σ over the pixels of the raw traces, once per file: every recompute's avg_fft follows from it by
    // linearity (see session_means) instead of from a pass over the spectra it has just written
    if (int rc = thz_pixel_sum(ctx, npix, s->nt, 1, s->d_raw, s->d_rawsum)) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return THZ_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// The recompute, in two halves so that a group of sessions (group_api.cpp: one x-slab per GPU) can put
// its exchange step between them:
//   session_enqueue  everything up to and including the fused launch, on the context's stream; leaves
//                    the slab's undivided sums in d_msum when the fast means are possible
//   session_means    pixel means from (possibly all-reduced) sums
// Chain positions (filter_chain of main.rs:182-247, "initial" = 0):
//   1 scaling  2 Tilt Compensation  3 Time Band Pass  4 fft  5 Frequency Band Pass [+ K14 / K13]
//   6 ifft  7 Time Band Pass (after)  8 Deconvolution
// ---------------------------------------------------------------------------------------------
namespace {

// fields of a chain configuration that decide everything in front of chain position 6
bool same_front(const thz_chain_cfg &a, const thz_chain_cfg &b)
{
    return a.tilt_active == b.tilt_active && a.tilt_x_deg == b.tilt_x_deg && a.tilt_y_deg == b.tilt_y_deg
           && a.td_before_active == b.td_before_active && a.td_before_low == b.td_before_low
           && a.td_before_high == b.td_before_high && a.td_before_width == b.td_before_width
           && a.fft_window.type == b.fft_window.type && a.fft_window.lower == b.fft_window.lower
           && a.fft_window.upper == b.fft_window.upper && a.fd_active == b.fd_active && a.fd_low == b.fd_low
           && a.fd_high == b.fd_high && a.fd_width == b.fd_width && a.scale_factor == b.scale_factor
           && a.want_means == b.want_means && a.avg_in_fourier_space == b.avg_in_fourier_space;
}

void post_multiplier(const thz_chain_cfg *cfg, const std::vector<float> &time, std::vector<float> &post)
{
    post.assign(time.size(), 1.0f);
    if (cfg->td_after_active) {
        double lo = cfg->td_after_low, hi = cfg->td_after_high;
        td_bandpass(time.data(), time.size(), &lo, &hi, cfg->td_after_width, post.data(), nullptr, nullptr);
    }
}

}  // namespace

// chain positions >= 6 on the resident band-passed spectrum: C2R, the new Time Band Pass, image
static int session_tail(thz_session *s, const thz_chain_cfg *cfg)
{
    thz_ctx *ctx = s->ctx;
    const size_t nt = s->nt_out, npix = s->nx_cur * s->ny_cur;
    if (ctx->time.size() != nt || std::memcmp(ctx->time.data(), s->time_out.data(), nt * sizeof(float)) != 0)
        if (int rc = thz_set_time_axis(ctx, s->time_out.data(), nt)) return rc;
    std::vector<float> post;
    post_multiplier(cfg, s->time_out, post);
    float *d_post = s->d_vec + nt;
    // through the session's pinned image of d_vec: asynchronous, and nobody writes h_vec again before this
    // recompute's closing synchronisation
    std::memcpy(s->h_vec + nt, post.data(), nt * sizeof(float));
    HIP_TRY(ctx, hipMemcpyAsync(d_post, s->h_vec + nt, nt * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    s->deconv_current = false;
    return thz_ifft(ctx, npix, s->d_fft, d_post, s->d_data, s->d_img);
}

SlabScale slab_scale(size_t nx_total, int world, int rank, size_t sf)
{
    SlabScale r;
    size_t x0 = 0, n = 0, acc = 0;
    const size_t nb = nx_total / sf;  // rows of the block grid (math_tools.rs:250: floor)
    for (int q = 0; q <= rank; ++q) {
        (void)thz_host_slab(nx_total, world, q, &x0, &n);
        SlabScale c;
        c.head = (sf - x0 % sf) % sf;
        if (c.head > n) { c.ok = false; c.head = n; }
        c.head_valid = c.head > 0 && x0 / sf < nb;
        const size_t b0 = (x0 + c.head) / sf;  // first block that starts inside the slab
        c.full = b0 < nb ? (n - c.head) / sf : 0;
        if (b0 + c.full > nb) c.full = nb - b0;
        const size_t rest = n - c.head - c.full * sf;
        c.tail = (b0 + c.full < nb) ? rest : 0;
        if (c.tail && q + 1 < world) {  // the next slab must hold the rest of that block
            size_t x1 = 0, n1 = 0;
            (void)thz_host_slab(nx_total, world, q + 1, &x1, &n1);
            if (n1 < sf - c.tail) c.ok = false;
        }
        c.rows = (c.head_valid ? 1 : 0) + c.full;
        c.x0 = acc;
        acc += c.rows;
        const bool ok = r.ok && c.ok;
        r = c;
        r.ok = ok;
    }
    return r;
}

int session_scale_tail(thz_session *s, const thz_chain_cfg *cfg)
{
    thz_ctx *ctx = s->ctx;
    s->carry_out_valid = false;
    const size_t sf = cfg->scale_factor > 1 ? (size_t)cfg->scale_factor : 1;
    if (sf <= 1 || !s->raw_grid_rows || s->ny / sf == 0 || s->raw_grid_rows / sf == 0) return THZ_OK;
    if (int rc = use_device(ctx)) return rc;
    const SlabScale sl = slab_scale(s->raw_grid_rows, s->slab_world, s->slab_rank, sf);
    if (!sl.ok) return fail(ctx, THZ_ERR_UNSUPPORTED, "scaling over slabs: a slab has fewer rows than the scale factor");
    const size_t need = (s->ny / sf) * s->nt;
    if (s->carry_floats != need) {
        s->carry_floats = 0;
        if (int rc = dev_alloc(ctx, &s->d_carry_out, need)) return rc;
        if (int rc = dev_alloc(ctx, &s->d_carry_in, need)) return rc;
        s->carry_floats = need;
    }
    if (sl.tail) {
        const float *rows = s->d_raw + (sl.head + sl.full * sf) * s->ny * s->nt;
        launch_scale_rows_partial(ctx->stream, rows, sl.tail, s->ny, s->nt, sf, nullptr, 0.0f, s->d_carry_out);
        if (int rc = check_launch(ctx)) return rc;
        s->carry_out_valid = true;
    }
    return THZ_OK;
}

int session_enqueue(thz_session *s, const thz_chain_cfg *cfg, int start_stage, bool *tail_only)
{
    thz_ctx *ctx = s->ctx;
    if (int rc = use_device(ctx)) return rc;
    *tail_only = false;
    if (start_stage >= 6 && s->have_outputs && s->have_last_cfg && same_front(s->last_cfg, *cfg)) {
        *tail_only = true;
        if (int rc = session_tail(s, cfg)) return rc;
        s->last_cfg = *cfg;
        return THZ_OK;
    }
    const float *src = s->d_raw;
    std::vector<float> time = s->time;
    std::vector<float> tilt_taper;
    size_t nt_cur = s->nt;
    bool tilt_as_multiplier = false, tilted = false;

    // ---- scaling (math_tools.rs:242-310): s x s block means of the raw cube, / s^2 also on ragged
    // edges; dx, dy grow by s; identity when s <= 1 or when a side would vanish (:244-256)
    size_t sf = cfg->scale_factor > 1 ? (size_t)cfg->scale_factor : 1;
    if (s->nx / sf == 0 || s->ny / sf == 0) sf = 1;
    // ---- Tilt Compensation, planned before any buffer is touched: a length no transform exists for
    // leaves the session as it was (the reference re-plans for any length)
    // a group's slab: the rows of the CURRENT grid it owns and where they sit in the whole grid (the Tilt plan and
    // the regions of interest depend on the position in the whole grid); behind a scaling stage, slab_scale's split
    SlabScale sl;
    const bool slab = s->raw_grid_rows != 0;
    if (slab && s->raw_grid_rows / sf == 0) sf = 1;
    if (slab && sf > 1) {
        sl = slab_scale(s->raw_grid_rows, s->slab_world, s->slab_rank, sf);
        if (!sl.ok) return fail(ctx, THZ_ERR_UNSUPPORTED, "scaling over slabs: a slab has fewer rows than the scale factor");
    }
    const size_t nx_c = slab && sf > 1 ? sl.rows : s->nx / sf, ny_c = s->ny / sf;
    const size_t g_rows = slab ? s->raw_grid_rows / sf : nx_c, g_x0 = slab ? (sf > 1 ? sl.x0 : s->raw_grid_x0) : 0;
    s->grid_rows = slab ? g_rows : 0;
    s->grid_x0 = g_x0;
    const float dx_c = s->dx * (float)sf, dy_c = s->dy * (float)sf;
    size_t steps = 0;
    if (cfg->tilt_active)
        steps = tilt_plan(time.data(), nt_cur, g_rows, ny_c, cfg->tilt_x_deg, cfg->tilt_y_deg, dx_c, dy_c, nullptr, nullptr);
    if (steps) {
        PlanHost probe;
        if (!build_plan(nt_cur + 2 * steps, probe, ctx->allow_f, ctx->allow_p))
            return fail(ctx, THZ_ERR_UNSUPPORTED,
                        "tilt compensation extends the traces to " + std::to_string(nt_cur + 2 * steps)
                            + " samples: no transform of that length (powers of two up to 16384, other lengths up to 8191)");
    }
    // what the fft stage reads changes with the scale factor and the tilt only
    if ((int)sf != s->last_sf || cfg->tilt_active != s->last_tilt_active
        || (cfg->tilt_active && (cfg->tilt_x_deg != s->last_tilt_x || cfg->tilt_y_deg != s->last_tilt_y))) {
        ++s->src_gen;
        s->last_sf = (int)sf; s->last_tilt_active = cfg->tilt_active; s->last_tilt_x = cfg->tilt_x_deg; s->last_tilt_y = cfg->tilt_y_deg;
    }
    s->have_outputs = false;  // the grid may change under the buffers below
    s->have_last_cfg = false;
    s->deconv_current = false;
    s->scale = sf;
    s->nx_cur = nx_c; s->ny_cur = ny_c;
    s->dx_cur = dx_c; s->dy_cur = dy_c;
    const size_t npix = s->nx_cur * s->ny_cur;
    if (sf > 1) {
        if (s->scaled_floats != npix * s->nt) {
            s->scaled_floats = 0;
            if (int rc = dev_alloc(ctx, &s->d_scaled, npix * s->nt)) return rc;
            s->scaled_floats = npix * s->nt;
        }
        if (slab) {
            // the block the previous slab started: its partial sums (d_carry_in, put there by the group) + this slab's
            // leading rows, divided; then the blocks wholly inside.  Same adds in the same order as one session's.
            float *o = s->d_scaled;
            if (sl.head_valid) {
                launch_scale_rows_partial(ctx->stream, s->d_raw, sl.head, s->ny, s->nt, sf, s->d_carry_in, (float)(sf * sf), o);
                if (int rc = check_launch(ctx)) return rc;
                o += ny_c * s->nt;
            }
            if (sl.full)
                if (int rc = thz_scale3d(ctx, s->d_raw + sl.head * s->ny * s->nt, sl.full * sf, s->ny, s->nt, 1, sf, o)) return rc;
        } else if (int rc = thz_scale3d(ctx, s->d_raw, s->nx, s->ny, s->nt, 1, sf, s->d_scaled)) return rc;
        src = s->d_scaled;
    }

    // ---- Tilt Compensation: zero tilt is just its tail taper (a multiplier); otherwise the
    // cube is re-laid out on an extended axis and the chain continues at the new length
    if (cfg->tilt_active) {
        tilt_taper.resize(nt_cur);
        adapted_blackman(time.data(), nt_cur, 0.0f, 7.0f, tilt_taper.data());
        if (steps == 0) {
            tilt_as_multiplier = true;
        } else {
            const size_t nt2 = nt_cur + 2 * steps;
            std::vector<float> new_time(nt2);
            std::vector<int32_t> ins(npix);
            if (slab) {  // the insert index depends on the position in the whole grid: plan it, keep this slab's rows
                std::vector<int32_t> all(g_rows * s->ny_cur);
                tilt_plan(time.data(), nt_cur, g_rows, s->ny_cur, cfg->tilt_x_deg, cfg->tilt_y_deg, s->dx_cur, s->dy_cur,
                          new_time.data(), all.data());
                std::copy(all.begin() + (long)(g_x0 * s->ny_cur), all.begin() + (long)((g_x0 + s->nx_cur) * s->ny_cur), ins.begin());
            } else {
                tilt_plan(time.data(), nt_cur, s->nx_cur, s->ny_cur, cfg->tilt_x_deg, cfg->tilt_y_deg, s->dx_cur, s->dy_cur,
                          new_time.data(), ins.data());
            }
            if (s->tilt_floats != npix * nt2) {
                s->tilt_floats = 0;
                if (int rc = dev_alloc(ctx, &s->d_tilt, npix * nt2)) return rc;
                s->tilt_floats = npix * nt2;
            }
            if (s->ins_count != npix) {
                s->ins_count = 0;
                if (int rc = dev_alloc(ctx, &s->d_ins, npix)) return rc;
                s->ins_count = npix;
            }
            if (int rc = alloc_outputs(s, nt2)) return rc;
            HIP_TRY(ctx, hipMemcpyAsync(s->d_ins, ins.data(), npix * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(s->d_vec, tilt_taper.data(), nt_cur * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
            launch_tilt(ctx->stream, npix, (int)nt_cur, (int)nt2, src, s->d_vec, s->d_ins, s->d_tilt);
            if (int rc = check_launch(ctx)) return rc;
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            src = s->d_tilt;
            time = new_time;
            nt_cur = nt2;
            tilted = true;
        }
    }
    if (int rc = alloc_outputs(s, nt_cur)) return rc;
    if (int rc = thz_set_time_axis(ctx, time.data(), nt_cur)) return rc;  // (re)plan, data_thread.rs:1194-1227
    s->time_out = time;
    const size_t nf = nt_cur / 2 + 1;
    if ((!s->fd_real.empty() && s->fd_real.size() != nf) || (!s->fd_cmask.empty() && s->fd_cmask.size() != 2 * nf))
        return fail(ctx, THZ_ERR_INVALID, "thz_session_set_fd_filters: the multipliers were given for " + std::to_string(s->fd_real.empty() ? s->fd_cmask.size() / 2 : s->fd_real.size())
                                              + " bins, this chain's spectra have " + std::to_string(nf));

    // ---- multipliers in the reference's f32 order: ((tilt * td_before) * fft_window)
    std::vector<float> pre(nt_cur, 1.0f), w(nt_cur), post, mask(nf, 1.0f);
    // ... and one by one for the regions of interest's reference-order roi_data (session_roi.cpp): the mean of the
    // fft stage's `data`, on which the reference performs the three multiplies in turn
    const bool want_sep = cfg->want_means == 2 && !s->rois.empty();
    std::vector<float> sep(want_sep ? 3 * nt_cur : 0);
    s->wsep_on[0] = s->wsep_on[1] = s->wsep_on[2] = false;
    if (tilt_as_multiplier) {
        pre = tilt_taper;
        if (want_sep) { std::memcpy(sep.data(), tilt_taper.data(), nt_cur * sizeof(float)); s->wsep_on[0] = true; }
    }
    if (cfg->td_before_active) {
        double lo = cfg->td_before_low, hi = cfg->td_before_high;
        td_bandpass(time.data(), nt_cur, &lo, &hi, cfg->td_before_width, w.data(), nullptr, nullptr);
        for (size_t i = 0; i < nt_cur; ++i) pre[i] = pre[i] * w[i];
        if (want_sep) { std::memcpy(sep.data() + nt_cur, w.data(), nt_cur * sizeof(float)); s->wsep_on[1] = true; }
    }
    fft_window(cfg->fft_window.type, time.data(), nt_cur, cfg->fft_window.lower, cfg->fft_window.upper, w.data());
    for (size_t i = 0; i < nt_cur; ++i) pre[i] = pre[i] * w[i];
    if (want_sep) {
        std::memcpy(sep.data() + 2 * nt_cur, w.data(), nt_cur * sizeof(float));
        s->wsep_on[2] = true;
        if (s->wsep_floats != 3 * nt_cur) {
            s->wsep_floats = 0;
            if (int rc = dev_alloc(ctx, &s->d_wsep, 3 * nt_cur)) return rc;
            s->wsep_floats = 3 * nt_cur;
        }
        if (int rc = thz_memcpy_h2d(ctx, s->d_wsep, sep.data(), sep.size() * sizeof(float))) return rc;
    }
    int64_t band_lo = 0, band_hi = 0;  // [lower, upper) of the Frequency Band Pass: zero outside (band_pass_fd.rs:194-212)
    if (cfg->fd_active)
        fd_bandpass(ctx->freq.data(), nf, cfg->fd_low, cfg->fd_high, cfg->fd_width, mask.data(), &band_lo, &band_hi);
    // further Frequency-domain plugins behind the band pass (K14: a real multiplier, one f32 multiply per
    // plugin like the band pass itself)
    if (!s->fd_real.empty())
        for (size_t k = 0; k < nf; ++k) mask[k] = mask[k] * s->fd_real[k];
    post_multiplier(cfg, time, post);
    float *d_pre = s->d_vec, *d_post = s->d_vec + nt_cur;
    // keep the masks 16-byte aligned for the kernels' vector reads
    float *d_mask = s->d_vec + ((2 * nt_cur + 3) & ~(size_t)3);
    float *d_cmask = d_mask + ((nf + 3) & ~(size_t)3);
    {
        // one asynchronous copy out of the session's pinned image of d_vec (four pageable copies and a stream
        // synchronisation cost 0.1 ms per recompute, 5 % of a 128-row slab's); h_vec is not written again before
        // this recompute's closing synchronisation
        float *h = s->h_vec;
        std::memcpy(h + (d_pre - s->d_vec), pre.data(), nt_cur * sizeof(float));
        std::memcpy(h + (d_post - s->d_vec), post.data(), nt_cur * sizeof(float));
        std::memcpy(h + (d_mask - s->d_vec), mask.data(), nf * sizeof(float));
        size_t used = (size_t)(d_mask - s->d_vec) + nf;
        if (!s->fd_cmask.empty()) {
            std::memcpy(h + (d_cmask - s->d_vec), s->fd_cmask.data(), 2 * nf * sizeof(float));
            used = (size_t)(d_cmask - s->d_vec) + 2 * nf;
        }
        HIP_TRY(ctx, hipMemcpyAsync(s->d_vec, h, used * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    }

    // ---- pixel means.  want_means == 1: amplitude / phase sums accumulated inside the fused launch, and
    // avg_fft by linearity — every multiplier in front of the transform is the same for all pixels unless
    // the cube was tilted, so mean_p(H FFT(w x_p)) = H FFT(w mean_p(x_p)): ONE extra transform instead of a
    // pass over the spectra.  want_means == 2 (and any tilted cube): the reference's summation order
    // (ndarray mean_axis twice, math_tools.rs:421-440), bit for bit, as three passes over the outputs.
    s->msum_fast = cfg->want_means == 1 && !tilted;
    // a group's slab of a tilted cube: sums that add up over the slabs — three passes over the outputs, any order
    s->msum_passes = cfg->want_means == 1 && tilted && slab;
    if (s->msum_fast || s->msum_passes) {
        if (s->msum_floats != nt_cur + 4 * nf) {
            s->msum_floats = 0;
            if (int rc = dev_alloc(ctx, &s->d_msum, nt_cur + 4 * nf)) return rc;
            s->msum_floats = nt_cur + 4 * nf;
        }
    }
    if (s->msum_fast) {
        // Σ of the source traces: cached at upload for the raw cube, one small pass for a block-averaged one
        if (src == s->d_raw) HIP_TRY(ctx, hipMemcpyAsync(s->d_msum, s->d_rawsum, nt_cur * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        else if (int rc = thz_pixel_sum(ctx, npix, nt_cur, 1, src, s->d_msum)) return rc;
    }
    thz_pipeline_io io{};
    io.d_raw = src; io.d_pre_win = d_pre; io.d_fd_mask = d_mask; io.d_fd_cmask = s->fd_cmask.empty() ? nullptr : d_cmask;
    io.d_post_win = d_post; io.d_fft = s->d_fft; io.d_amp = s->d_amp; io.d_phase = s->d_ph; io.d_data_out = s->d_data;
    io.d_img = s->d_img; io.d_sums = s->msum_fast ? s->d_msum + nt_cur : nullptr;
    io.band_lo = io.band_hi = 0;
    // the band pass's own index range tells the fused kernel where its multiplier is zero (a further real plugin, K14,
    // only adds zeros inside)
    const bool band_known = cfg->fd_active && band_hi > band_lo && band_lo >= 0 && (size_t)band_hi <= nf;
    if (int rc = pipeline_ex_band(ctx, npix, &io, band_known ? (size_t)band_lo : 0, band_known ? (size_t)band_hi : 0)) return rc;
    if (s->msum_passes) {
        if (int rc = thz_pixel_sum(ctx, npix, nf, 2, s->d_fft, s->d_msum + nt_cur)) return rc;
        if (int rc = thz_pixel_sum(ctx, npix, nf, 1, s->d_amp, s->d_msum + nt_cur + 2 * nf)) return rc;
        if (int rc = thz_pixel_sum(ctx, npix, nf, 1, s->d_ph, s->d_msum + nt_cur + 3 * nf)) return rc;
    }
    s->have_means = false;
    s->have_outputs = true;
    s->deconv_current = false;  // the stage passes its input through unless it is the one updated
    s->d_src = src;
    s->last_cfg = *cfg;
    s->have_last_cfg = true;
    return THZ_OK;
}

// Pixel means of the ifft stage.  total_pix: pixels of the WHOLE grid the sums in d_msum now cover (the
// slab's own, or — after the group's all-reduce — all slabs').
int session_means(thz_session *s, const thz_chain_cfg *cfg, size_t total_pix)
{
    thz_ctx *ctx = s->ctx;
    if (int rc = use_device(ctx)) return rc;
    const size_t nt = s->nt_out, nf = s->nf_out;
    if (!cfg->want_means) return THZ_OK;
    if (s->msum_passes) {
        launch_scale_vec(ctx->stream, s->d_msum + nt, 1.0f / (float)total_pix, 4 * nf, s->d_avg);
        if (int rc = check_launch(ctx)) return rc;
    } else if (s->msum_fast) {
        const float inv = 1.0f / (float)total_pix;
        // amplitudes and phases: sums / pixels; mean source trace in place
        launch_scale_vec(ctx->stream, s->d_msum + nt, inv, 2 * nf, s->d_avg + 2 * nf);
        launch_scale_vec(ctx->stream, s->d_msum, inv, nt, s->d_msum);
        if (int rc = check_launch(ctx)) return rc;
        // avg_fft = (cmask mask) FFT(pre * mean trace): the same kernels as the cube's own transform
        float *d_pre = s->d_vec;
        float *d_mask = s->d_vec + ((2 * nt + 3) & ~(size_t)3);
        float *d_cmask = d_mask + ((nf + 3) & ~(size_t)3);
        if (int rc = thz_fft(ctx, 1, s->d_msum, d_pre, nullptr, nullptr, s->d_avg, nullptr, nullptr, d_mask)) return rc;
        if (!s->fd_cmask.empty())
            if (int rc = thz_apply_fd_cmask(ctx, 1, s->d_avg, nullptr, d_cmask)) return rc;
    } else {
        if (int rc = thz_pixel_mean(ctx, s->nx_cur, s->ny_cur, nf, 2, s->d_fft, s->d_avg)) return rc;
        if (int rc = thz_pixel_mean(ctx, s->nx_cur, s->ny_cur, nf, 1, s->d_amp, s->d_avg + 2 * nf)) return rc;
        if (int rc = thz_pixel_mean(ctx, s->nx_cur, s->ny_cur, nf, 1, s->d_ph, s->d_avg + 3 * nf)) return rc;
    }
    s->have_means = true;
    return THZ_OK;
}

extern "C" {

int thz_session_recompute_from(thz_session *s, const thz_chain_cfg *cfg, int start_stage)
{
    if (!s || !cfg) return THZ_ERR_INVALID;
    thz_ctx *ctx = s->ctx;
    if (start_stage < 0 || start_stage > 8) return fail(ctx, THZ_ERR_INVALID, "thz_session_recompute_from: chain positions are 1..8");
    if (start_stage == 8) return THZ_OK;  // only the Deconvolution stage is re-run: thz_session_deconvolve
    bool tail_only = false;
    if (int rc = session_enqueue(s, cfg, start_stage, &tail_only)) return rc;
    if (!tail_only) {
        if (int rc = session_means(s, cfg, s->nx_cur * s->ny_cur)) return rc;
        if (int rc = session_avg_data(s, cfg)) return rc;
    }
    // regions of interest: the ifft stage's per-region means + the plot copy-out's (session_roi.cpp)
    bool roi_data_only = tail_only;
    if (int rc = session_roi_sums(s, cfg, &roi_data_only)) return rc;
    if (int rc = session_roi_finish(s, cfg, roi_data_only)) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return THZ_OK;
}

int thz_session_recompute(thz_session *s, const thz_chain_cfg *cfg) { return thz_session_recompute_from(s, cfg, 1); }

int thz_session_set_fd_filters(thz_session *s, const float *real_mask, const float *cmask, size_t nf)
{
    if (!s) return THZ_ERR_INVALID;
    if ((real_mask || cmask) && nf == 0) return fail(s->ctx, THZ_ERR_INVALID, "thz_session_set_fd_filters: nf is 0");
    s->fd_real.assign(real_mask ? real_mask : nullptr, real_mask ? real_mask + nf : nullptr);
    s->fd_cmask.assign(cmask ? cmask : nullptr, cmask ? cmask + 2 * nf : nullptr);
    s->have_last_cfg = false;  // the next recompute starts at the front whatever its start position says
    return THZ_OK;
}

}  // extern "C"

extern "C" {

int thz_session_deconvolve(thz_session *s, const thz_psf *psf, const thz_deconv_cfg *cfg,
                           volatile const int *abort_flag, float *progress)
{
    if (!s || !psf || !cfg) return THZ_ERR_INVALID;
    thz_ctx *ctx = s->ctx;
    if (!s->have_outputs) return fail(ctx, THZ_ERR_NOT_READY, "thz_session_deconvolve: no recompute has run");
    if (int rc = use_device(ctx)) return rc;
    const size_t npix = s->nx_cur * s->ny_cur, n = npix * s->nt_out;
    if (s->deconv_floats != n) {
        s->deconv_floats = 0;
        s->deconv_current = false;
        if (int rc = dev_alloc(ctx, &s->d_deconv, n)) return rc;
        if (int rc = dev_alloc(ctx, &s->d_deconv_img, npix)) return rc;
        s->deconv_floats = n;
    }
    // the engine's axis is the chain's current one unless a plot call re-planned in between
    if (ctx->time.size() != s->nt_out)
        if (int rc = thz_set_time_axis(ctx, s->time_out.data(), s->nt_out)) return rc;
    // the stage's input is always the Time Band Pass output, never its own earlier result
    const int rc = thz_deconvolve(ctx, psf, cfg, s->nx_cur, s->ny_cur, s->dx_cur, s->dy_cur, s->d_data, s->d_deconv, s->d_deconv_img,
                                  nullptr, abort_flag, progress);
    if (rc < 0) {  // aborted or failed: filter() hands back input.clone()
        s->deconv_current = false;
        return rc;
    }
    s->deconv_current = true;  // THZ_SKIPPED too: the guards copied the input through
    if (s->have_last_cfg && !s->rois.empty()) {  // the regions' means of the FINAL traces follow the stage's output
        bool roi_data_only = true;
        if (int rc2 = session_roi_sums(s, &s->last_cfg, &roi_data_only)) return rc2;
        if (int rc2 = session_roi_finish(s, &s->last_cfg, roi_data_only)) return rc2;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return rc;
}

size_t thz_session_nt_out(const thz_session *s) { return s ? s->nt_out : 0; }

int thz_session_grid(const thz_session *s, size_t *nx, size_t *ny, float *dx, float *dy)
{
    if (!s) return THZ_ERR_INVALID;
    if (nx) *nx = s->nx_cur;
    if (ny) *ny = s->ny_cur;
    if (dx) *dx = s->dx_cur;
    if (dy) *dy = s->dy_cur;
    return THZ_OK;
}

int thz_session_time_out(const thz_session *s, float *time)
{
    if (!s || !time) return THZ_ERR_INVALID;
    std::memcpy(time, s->time_out.data(), s->time_out.size() * sizeof(float));
    return THZ_OK;
}

void *thz_session_buffer(thz_session *s, int which)
{
    if (!s) return nullptr;
    const size_t nf = s->nf_out;
    // Before the first recompute (and between an upload and the next recompute) the output buffers hold
    // nothing a caller may read — and after a scaled recompute they are sized for the smaller grid while
    // the session's grid is the raw one again: absent (NULL / THZ_ERR_NOT_READY) until a recompute has run.
    // The image is the exception: thz_session_upload fills it for the raw grid (io.rs:588-594).
    if (!s->have_outputs && which != THZ_BUF_RAW && which != THZ_BUF_IMG) return nullptr;
    switch (which) {
    case THZ_BUF_RAW: return s->d_raw;
    case THZ_BUF_FFT: return s->d_fft;
    case THZ_BUF_AMPLITUDES: return s->d_amp;
    case THZ_BUF_PHASES: return s->d_ph;
    case THZ_BUF_DATA: return final_data(s);
    case THZ_BUF_IMG: return final_img(s);
    case THZ_BUF_AVG_FFT: return s->have_means ? s->d_avg : nullptr;
    case THZ_BUF_AVG_AMPLITUDES: return s->have_means ? s->d_avg + 2 * nf : nullptr;
    case THZ_BUF_AVG_PHASES: return s->have_means ? s->d_avg + 3 * nf : nullptr;
    case THZ_BUF_OPACITY: return s->opacity_floats == s->nx_cur * s->ny_cur * s->nt_out ? s->d_opacity : nullptr;
    default: return nullptr;
    }
}

int thz_session_download(thz_session *s, int which, size_t pix0, size_t npix, void *dst)
{
    if (!s || !dst) return THZ_ERR_INVALID;
    thz_ctx *ctx = s->ctx;
    const float *base = static_cast<const float *>(thz_session_buffer(s, which));
    if (!base) return fail(ctx, THZ_ERR_NOT_READY, "thz_session_download: buffer not available");
    size_t per = 0;  // floats per pixel
    // per-pixel outputs were allocated for out_pix pixels by the recompute that filled them; the image has
    // nx * ny entries (allocated once) of which the current grid's are valid
    const size_t total_pix = which == THZ_BUF_RAW ? s->nx * s->ny
                             : which == THZ_BUF_IMG ? s->nx_cur * s->ny_cur
                                                    : (s->out_pix < s->nx_cur * s->ny_cur ? s->out_pix : s->nx_cur * s->ny_cur);
    switch (which) {
    case THZ_BUF_RAW: per = s->nt; break;
    case THZ_BUF_FFT: per = 2 * s->nf_out; break;
    case THZ_BUF_AMPLITUDES: case THZ_BUF_PHASES: per = s->nf_out; break;
    case THZ_BUF_DATA: case THZ_BUF_OPACITY: per = s->nt_out; break;
    case THZ_BUF_IMG: per = 1; break;
    case THZ_BUF_AVG_FFT: return thz_memcpy_d2h(ctx, dst, base, 2 * s->nf_out * sizeof(float));
    case THZ_BUF_AVG_AMPLITUDES: case THZ_BUF_AVG_PHASES: return thz_memcpy_d2h(ctx, dst, base, s->nf_out * sizeof(float));
    default: return THZ_ERR_INVALID;
    }
    if (pix0 > total_pix || npix > total_pix - pix0) return fail(ctx, THZ_ERR_INVALID, "thz_session_download: pixel range out of bounds");
    return thz_memcpy_d2h(ctx, dst, base + pix0 * per, npix * per * sizeof(float));
}

int thz_session_voxels(thz_session *s, const thz_voxel_cfg *cfg, uint64_t max_instances, int scaling,
                       size_t orig_w, size_t orig_h, size_t orig_d, thz_voxel_instance *host_out,
                       uint64_t capacity, uint64_t *count, float *threshold, float *cube_dims)
{
    if (!s || !cfg || !count) return THZ_ERR_INVALID;
    thz_ctx *ctx = s->ctx;
    if (!s->have_outputs) return fail(ctx, THZ_ERR_NOT_READY, "thz_session_voxels: no recompute has run");
    if (capacity && !host_out) return fail(ctx, THZ_ERR_INVALID, "thz_session_voxels: capacity without a buffer");
    if (int rc = use_device(ctx)) return rc;
    const size_t npix = s->nx_cur * s->ny_cur, nt = s->nt_out, n = npix * nt;
    if (s->opacity_floats != n) {
        s->opacity_floats = 0;
        if (int rc = dev_alloc(ctx, &s->d_opacity, n)) return rc;
        s->opacity_floats = n;
    }
    if (int rc = thz_voxel_opacity(ctx, npix, nt, final_data(s), cfg, s->d_opacity)) return rc;
    float thr = 0.0f;
    if (int rc = thz_voxel_threshold(ctx, s->d_opacity, n, max_instances, &thr)) return rc;
    if (threshold) *threshold = thr;
    const float time_span = s->time_out.back() - s->time_out.front();
    thz_voxel_instance *d_inst = nullptr;
    if (capacity) HIP_TRY(ctx, hipMalloc((void **)&d_inst, capacity * sizeof(thz_voxel_instance)));
    int rc = thz_voxel_instances(ctx, s->d_opacity, s->nx_cur, s->ny_cur, nt, 0, s->nx_cur, thr, time_span, scaling, orig_w, orig_h,
                                 orig_d, d_inst, capacity, count, cube_dims);
    if (!rc && capacity) {
        const uint64_t n_copy = *count < capacity ? *count : capacity;
        if (n_copy) rc = thz_memcpy_d2h(ctx, host_out, d_inst, n_copy * sizeof(thz_voxel_instance));
    }
    if (d_inst) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(d_inst);
    }
    return rc;
}

int thz_session_plot(thz_session *s, size_t px, size_t py, const thz_plot_out *out)
{
    if (!s || !out) return THZ_ERR_INVALID;
    thz_ctx *ctx = s->ctx;
    if (px >= s->nx || py >= s->ny) return fail(ctx, THZ_ERR_INVALID, "thz_session_plot: pixel out of bounds");
    if (int rc = use_device(ctx)) return rc;
    if (out->signal)
        if (int rc = thz_session_download(s, THZ_BUF_RAW, px * s->ny + py, 1, out->signal)) return rc;
    const bool need_outputs = out->signal_fft || out->phase_fft || out->filtered_signal || out->filtered_signal_fft
                              || out->filtered_phase_fft || out->avg_signal || out->avg_signal_fft || out->avg_phase_fft;
    if (!need_outputs) return THZ_OK;
    if (!s->have_outputs) return fail(ctx, THZ_ERR_NOT_READY, "thz_session_plot: no recompute has run");
    // everything behind the scaling stage lives on the block grid: pixel / scale (a ragged edge has no block)
    size_t row = px / s->scale;
    if (s->raw_grid_rows && s->scale > 1) {
        // a group's slab behind a scaling stage: the block of raw row px belongs to the slab that holds its LAST row
        const size_t block = (s->raw_grid_x0 + px) / s->scale;
        if (block < s->grid_x0 || block >= s->grid_x0 + s->nx_cur)
            return fail(ctx, THZ_ERR_INVALID, "thz_session_plot: this pixel's block is held by a neighbouring slab (or lies beyond the scaled grid)");
        row = block - s->grid_x0;
    }
    if (row >= s->nx_cur || py / s->scale >= s->ny_cur)
        return fail(ctx, THZ_ERR_INVALID, "thz_session_plot: pixel beyond the scaled grid");
    const size_t pix = row * s->ny_cur + py / s->scale;
    const size_t nt = s->nt_out, nf = s->nf_out;
    if (out->filtered_signal)
        if (int rc = thz_session_download(s, THZ_BUF_DATA, pix, 1, out->filtered_signal)) return rc;
    if (out->filtered_signal_fft)
        if (int rc = thz_session_download(s, THZ_BUF_AMPLITUDES, pix, 1, out->filtered_signal_fft)) return rc;
    if (out->filtered_phase_fft)
        if (int rc = thz_session_download(s, THZ_BUF_PHASES, pix, 1, out->filtered_phase_fft)) return rc;
    if (out->avg_signal_fft)
        if (int rc = thz_session_download(s, THZ_BUF_AVG_AMPLITUDES, 0, 1, out->avg_signal_fft)) return rc;
    if (out->avg_phase_fft)
        if (int rc = thz_session_download(s, THZ_BUF_AVG_PHASES, 0, 1, out->avg_phase_fft)) return rc;
    if (out->signal_fft || out->phase_fft || out->avg_signal) {
        // scratch behind the multipliers: [amp nf | phase nf | mean nt]
        float *d_tmp = nullptr;
        HIP_TRY(ctx, hipMalloc((void **)&d_tmp, (2 * nf + nt) * sizeof(float)));
        int rc = THZ_OK;
        if (out->signal_fft || out->phase_fft) {
            // the fft stage's own amplitudes / phases (no band-pass yet): one trace through K1-K3 again
            if (ctx->time.size() != nt) rc = thz_set_time_axis(ctx, s->time_out.data(), nt);
            const float *src = s->d_src + pix * nt;
            if (!rc) rc = thz_fft(ctx, 1, src, s->d_vec /* pre */, nullptr, nullptr, nullptr, d_tmp, d_tmp + nf, nullptr);
            if (!rc && out->signal_fft) rc = thz_memcpy_d2h(ctx, out->signal_fft, d_tmp, nf * sizeof(float));
            if (!rc && out->phase_fft) rc = thz_memcpy_d2h(ctx, out->phase_fft, d_tmp + nf, nf * sizeof(float));
        }
        if (!rc && out->avg_signal) {
            if (s->have_last_cfg && s->last_cfg.avg_in_fourier_space && s->avg_data.size() == nt) {
                std::memcpy(out->avg_signal, s->avg_data.data(), nt * sizeof(float));  // filtered.avg_data, data_thread.rs:1431
            } else {
                rc = thz_pixel_mean(ctx, s->nx_cur, s->ny_cur, nt, 1, final_data(s), d_tmp + 2 * nf);
                if (!rc) rc = thz_memcpy_d2h(ctx, out->avg_signal, d_tmp + 2 * nf, nt * sizeof(float));
            }
        }
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(d_tmp);
        if (rc) return rc;
    }
    return THZ_OK;
}

}  // extern "C"
