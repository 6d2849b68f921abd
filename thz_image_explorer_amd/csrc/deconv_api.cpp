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

using namespace thz;

namespace {

struct DevFree {
    std::vector<void *> ptrs;
    ~DevFree()
    {
        for (void *p : ptrs) (void)hipFree(p);
    }
    template <class T>
    hipError_t alloc(T **out, size_t bytes)
    {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
        if (e == hipSuccess) ptrs.push_back(p);
        *out = reinterpret_cast<T *>(p);
        return e;
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

int thz_deconvolve(thz_ctx *ctx, const thz_psf *psf, const thz_deconv_cfg *cfg, size_t nx, size_t ny,
                   float dx, float dy, const float *d_in, float *d_out, float *d_img,
                   float *d_gains_out, volatile const int *abort_flag, float *progress)
{
    if (int rc = need_plan(ctx)) return rc;
    if (!psf || !cfg || !d_in || !d_out || nx == 0 || ny == 0)
        return fail(ctx, THZ_ERR_INVALID, "thz_deconvolve: bad argument");
    const size_t nt = ctx->time.size(), npix = nx * ny;
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
        filter_bank(nb, (double)cfg->start_freq, (double)cfg->end_freq, (double)cfg->win_width,
                    ctx->time.data(), filters, centers);
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
    if (b0 == 0 && b1 == 0) b1 = nb;
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
        if (int rc = pass_through()) return rc;
        if (progress) *progress = 1.0f;
        return THZ_SKIPPED;
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
    PlanHost H;
    const c32 *d_f[3] = {nullptr, nullptr, nullptr};
    if (M > 16384 || !build_plan(M, H, true))  // with the F core's tables where M has them (band energies)
        return fail(ctx, THZ_ERR_UNSUPPORTED, "thz_deconvolve: trace too long for the FIR transform");
    const size_t N = M / 2, nk = N + 1;
    DevFree mem;
    c32 *d_tw = nullptr, *d_spec = nullptr, *d_H = nullptr;
    float *d_energy = nullptr, *d_gain = nullptr, *d_ws = nullptr;
    RlBand *d_bands = nullptr;
    {
        std::vector<c32> pack(H.tw);
        pack.insert(pack.end(), H.tw_split.begin(), H.tw_split.end());
        const bool f = H.family == kFamilyF && !H.f_t1.empty();
        const size_t o1 = pack.size();
        if (f) pack.insert(pack.end(), H.f_t1.begin(), H.f_t1.end());
        const size_t o2 = pack.size();
        if (f) pack.insert(pack.end(), H.f_t2.begin(), H.f_t2.end());
        const size_t o3 = pack.size();
        if (f) pack.insert(pack.end(), H.f_w2n.begin(), H.f_w2n.end());
        HIP_TRY(ctx, mem.alloc(&d_tw, pack.size() * sizeof(c32)));
        HIP_TRY(ctx, hipMemcpyAsync(d_tw, pack.data(), pack.size() * sizeof(c32), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // pack goes out of scope
        H.f_t1.clear();  // offsets only from here on
        H.f_t2.clear();
        H.f_w2n.clear();
        // the transform kernels of this call are the generic (LDS) ones; only the tables ride along
        d_f[0] = f ? d_tw + o1 : nullptr;
        d_f[1] = f ? d_tw + o2 : nullptr;
        d_f[2] = f ? d_tw + o3 : nullptr;
    }
    PlanDev P = plan_dev(H, d_tw, d_tw + H.tw.size(), nullptr, nullptr);
    P.f_t1 = d_f[0];
    P.f_t2 = d_f[1];
    P.f_w2n = d_f[2];
    tick("plan, twiddles");
    // ---- filter spectra H_b[k] = (1/M) sum_j h_b[j] exp(-2 pi i j k / M), in double, for the bands of
    // this call (band-parallel multi-GPU: cfg->band_begin/band_end select a subset of the bank)
    const int nbs = b1 - b0;
    {
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
        HIP_TRY(ctx, mem.alloc(&d_H, (size_t)nbs * nk * sizeof(c32)));
        HIP_TRY(ctx, hipMemcpyAsync(d_trig, trig.data(), trig.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_filters, filters.data() + (size_t)b0 * kDeconvTaps,
                                    (size_t)nbs * kDeconvTaps * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        launch_dc_filter_spectra(ctx->stream, d_filters, nbs, kDeconvTaps, d_trig, d_trig + M, (unsigned)M,
                                 (unsigned)nk, d_H);
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // trig goes out of scope
    }
    centers = std::vector<float>(centers.begin() + b0, centers.begin() + b1);
    tick("filter spectra");
    // ---- per-band PSFs, iteration counts, workspace layout
    std::vector<RlBand> bands((size_t)nbs);
    std::vector<float> psf_pack;
    size_t ws_floats = 0;
    unsigned blk = 0, tblk = 0;
    size_t tile_lds = 0;
    int max_iter = 0;
    std::vector<BandPsf> psfs((size_t)nbs);
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
        B.tiles_w = (B.W + 15) / 16;
        B.n_tiles = B.tiles_w * ((B.H + 15) / 16);
        tile_lds = std::max(tile_lds, rl_tile_lds_bytes(B.pr, B.pc));
        const size_t img = (size_t)B.H * B.W;
        B.off_d = (unsigned)ws_floats; ws_floats += img;
        B.off_u = (unsigned)ws_floats; ws_floats += img;
        B.off_t = (unsigned)ws_floats; ws_floats += img;
    }
    // The tiled grid lists the bands by falling iteration count: at iteration `it` the tiles that still
    // iterate are a PREFIX of the grid (live_blocks(it)), so a launch need not carry the blocks of finished bands —
    // late in the call that is the difference between ~100 blocks and ~1 500 that each cost a dispatch only to
    // find out they have nothing to do.
    std::vector<int> order((size_t)nbs);
    for (int b = 0; b < nbs; ++b) order[(size_t)b] = b;
    std::stable_sort(order.begin(), order.end(), [&](int a, int c) { return bands[(size_t)a].n_iter > bands[(size_t)c].n_iter; });
    std::vector<std::pair<int, unsigned>> live_steps;  // (n_iter of a band, blocks up to and including it), by falling n_iter
    for (int o = 0; o < nbs; ++o) {
        RlBand &B = bands[(size_t)order[(size_t)o]];
        B.tblk0 = tblk;
        tblk += rl_tile_block_count(B.pr, B.pc, (unsigned)B.n_tiles);
        live_steps.emplace_back(B.n_iter, tblk);
    }
    auto live_blocks = [&](int it) {  // blocks of the bands with n_iter > it
        unsigned n = 0;
        for (const auto &st : live_steps)
            if (st.first > it) n = st.second;
        return n;
    };
    for (int b = 0; b < nbs; ++b) {
        const BandPsf &bp = psfs[(size_t)b];
        RlBand &B = bands[(size_t)b];
        B.off_psf = (unsigned)(ws_floats + psf_pack.size());
        psf_pack.insert(psf_pack.end(), bp.v.begin(), bp.v.end());
        B.off_mirror = (unsigned)(ws_floats + psf_pack.size());
        for (size_t i = bp.v.size(); i-- > 0;) psf_pack.push_back(bp.v[i]);  // psf[::-1, ::-1]
    }
    if (getenv("THZ_DEBUG_BANDS"))  // developer knob: the band table on stderr
        for (int b = 0; b < nbs; ++b)
            fprintf(stderr, "band %2d  f=%.3f THz  psf %3d x %3d  n_iter %4d  tiles %u\n", b, centers[(size_t)b],
                    bands[(size_t)b].pr, bands[(size_t)b].pc, bands[(size_t)b].n_iter,
                    rl_tile_block_count(bands[(size_t)b].pr, bands[(size_t)b].pc, (unsigned)bands[(size_t)b].n_tiles));
    psf_pack.insert(psf_pack.end(), 32, 0.0f);  // the tiled step reads taps a whole chunk at a time
    HIP_TRY(ctx, mem.alloc(&d_ws, (ws_floats + psf_pack.size()) * sizeof(float)));
    HIP_TRY(ctx, hipMemcpyAsync(d_ws + ws_floats, psf_pack.data(), psf_pack.size() * sizeof(float),
                                hipMemcpyHostToDevice, ctx->stream));
    std::vector<RlTileRef> tiles(tblk);
    for (int o = 0; o < nbs; ++o) {
        const RlBand &B = bands[(size_t)order[(size_t)o]];
        const unsigned end = o + 1 < nbs ? bands[(size_t)order[(size_t)o + 1]].tblk0 : tblk;
        for (unsigned t = B.tblk0; t < end; ++t) tiles[t] = RlTileRef{B};
    }
    RlTileRef *d_tiles = nullptr;
    HIP_TRY(ctx, mem.alloc(&d_tiles, tiles.size() * sizeof(RlTileRef)));
    HIP_TRY(ctx, hipMemcpyAsync(d_tiles, tiles.data(), tiles.size() * sizeof(RlTileRef), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, mem.alloc(&d_bands, bands.size() * sizeof(RlBand)));
    HIP_TRY(ctx, hipMemcpyAsync(d_bands, bands.data(), bands.size() * sizeof(RlBand), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, mem.alloc(&d_spec, npix * nk * sizeof(c32)));
    HIP_TRY(ctx, mem.alloc(&d_energy, (size_t)nbs * npix * sizeof(float)));
    HIP_TRY(ctx, mem.alloc(&d_gain, (size_t)nbs * npix * sizeof(float)));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // host staging vectors go out of scope below

    tick("band PSFs, workspace");
    const int shift = (kDeconvTaps - 1) / 2;
    launch_dc_fft(ctx->stream, P, npix, (int)nt, d_in, d_spec);
    launch_dc_energy(ctx->stream, P, npix, (int)nt, nbs, shift, d_spec, d_H, d_energy);
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
    if (tiled) prepare_rl_step_tiled(tile_lds);
    auto enqueue = [&](const int *it_base, int it, unsigned grid_blocks) {
        if (tiled) {
            if (grid_blocks == 0) return;
            launch_rl_step_tiled(ctx->stream, d_tiles, grid_blocks, tile_lds, it_base, it, 0, d_ws);
            launch_rl_step_tiled(ctx->stream, d_tiles, grid_blocks, tile_lds, it_base, it, 1, d_ws);
        } else {
            launch_rl_step(ctx->stream, d_bands, nbs, blk, it_base, it, 0, d_ws);
            launch_rl_step(ctx->stream, d_bands, nbs, blk, it_base, it, 1, d_ws);
        }
    };
    int *d_it = nullptr;
    HIP_TRY(ctx, mem.alloc(&d_it, sizeof(int)));
    const bool use_graph = !getenv("THZ_NO_GRAPH");      // developer knobs: plain launches / the whole grid
    const bool compact = !getenv("THZ_RL_FULL_GRID");    // every time, for A/B timing
    // One captured batch per grid size: a batch that starts at iteration `base` launches the live prefix of the
    // tile grid, rounded up to a power of two so that a call needs a handful of graphs, not one per band.
    struct GraphCache {
        std::vector<std::pair<unsigned, std::pair<hipGraph_t, hipGraphExec_t>>> g;
        ~GraphCache()
        {
            for (auto &e : g) {
                if (e.second.second) (void)hipGraphExecDestroy(e.second.second);
                if (e.second.first) (void)hipGraphDestroy(e.second.first);
            }
        }
    } cache;
    auto grid_for = [&](int base) -> unsigned {
        if (!tiled) return blk;
        if (!compact) return tblk;
        const unsigned live = live_blocks(base);
        unsigned g2 = 1;
        while (g2 < live) g2 <<= 1;
        return g2 < tblk ? g2 : tblk;
    };
    auto graph_for = [&](unsigned grid) -> hipGraphExec_t {
        for (auto &e : cache.g)
            if (e.first == grid) return e.second.second;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            for (int o = 0; o < kRlBatch; ++o) enqueue(d_it, o, grid);
            if (hipStreamEndCapture(ctx->stream, &graph) != hipSuccess || !graph
                || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
                if (graph) (void)hipGraphDestroy(graph);
                graph = nullptr;
                exec = nullptr;
                (void)hipGetLastError();
            }
        }
        cache.g.push_back({grid, {graph, exec}});
        return exec;
    };
    for (int base = 0; base < max_iter; base += kRlBatch) {
        if (abort_flag && *abort_flag) {  // cancellable_loops semantics: polled between batches
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (int rc = pass_through()) return rc;
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            return fail(ctx, THZ_ERR_ABORTED, "thz_deconvolve: aborted");
        }
        const unsigned grid = grid_for(base);
        hipGraphExec_t exec = (use_graph && max_iter > kRlBatch) ? graph_for(grid) : nullptr;
        if (exec) {
            HIP_TRY(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_it), base, 1, ctx->stream));
            HIP_TRY(ctx, hipGraphLaunch(exec, ctx->stream));
        } else {
            for (int it = base; it < base + kRlBatch && it < max_iter; ++it) enqueue(nullptr, it, compact && tiled ? live_blocks(it) : grid);
        }
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // keeps the abort poll honest
        if (progress) *progress = (float)std::min(base + kRlBatch, max_iter) / (float)max_iter;
    }
    tick("iterations");
    launch_dc_gain(ctx->stream, d_bands, nbs, npix, d_energy, d_ws, d_gain);
    launch_dc_combine(ctx->stream, P, npix, (int)nt, nbs, shift, d_spec, d_H, d_gain, d_out, d_img);
    if (int rc = check_launch(ctx)) return rc;
    if (d_gains_out)
        HIP_TRY(ctx, hipMemcpyAsync(d_gains_out, d_gain, (size_t)nbs * npix * sizeof(float),
                                    hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // temporaries are freed on return
    tick("gains, recombination");
    if (progress) *progress = 1.0f;
    return THZ_OK;
}

}  // extern "C"
