// ctx.hpp — the context object behind thz_ctx* and small helpers shared by the
// C-ABI translation units (api.cpp, deconv_api.cpp).
#pragma once
#include "../../include/thzgpu.h"

#include "kernels.hpp"
#include "plan_host.hpp"

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

using namespace thz;

struct thz_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    std::vector<float> time, freq;
    PlanHost plan_h;
    PlanDev plan_d{};
    c32 *d_tables = nullptr;  // one allocation: tw | tw_split | chirp_conj | bfft
    c32 *d_big = nullptr;     // global scratch of a plan whose buffers do not fit LDS (plan_d.big_scratch), or null
    bool have_plan = false;
    bool allow_f = true, allow_p = true;  // thz_set_kernel_family
    bool plan_allow_f = true, plan_allow_p = true;  // the switches the current plan was built under
    hipStream_t aux_streams[3] = {nullptr, nullptr, nullptr};  // the deconvolution's extra chains (created on first use)
    void *ws = nullptr;  // scratch workspace (pixel means, ROI lists)
    size_t ws_bytes = 0;
    // device blocks of the last thz_deconvolve call, kept for the next one: a call of the same geometry then neither
    // allocates nor frees (eleven hipMalloc + hipFree pairs measured 0.8 ms of a 10 ms call); blocks a call did
    // not use are freed at its end, so a change of geometry does not accumulate memory
    struct Block {
        void *p;
        size_t bytes;
        bool in_use, used_this_call;
    };
    std::vector<Block> dc_pool;
    // ... and the iteration-batch graph of each of its chains, valid for as long as the next call launches the same
    // kernel over the same grid with the same pointers (which the pool makes the normal case): capture and
    // instantiation of two 64-node graphs cost 0.2-0.3 ms of a 9 ms call
    struct ChainGraph {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        int kind = -1;
        unsigned blocks = 0;
        size_t lds = 0;
        const void *tiles = nullptr, *it = nullptr, *ws = nullptr;
        void drop()
        {
            if (exec) (void)hipGraphExecDestroy(exec);
            if (graph) (void)hipGraphDestroy(graph);
            exec = nullptr;
            graph = nullptr;
            kind = -1;
        }
    };
    ChainGraph dc_graph[4];
    // ... and what the call derives from the time axis and the configuration alone: the FIR bank (host, 0.24 ms),
    // the transform tables of the padded length (0.08 ms) and the bands' filter spectra (0.15 ms), each recomputed
    // only when what it depends on has changed
    struct DcBank {
        std::vector<float> time, filters, centers;
        int n_filters = -1;
        float start_freq = 0.0f, end_freq = 0.0f, win_width = 0.0f;
        unsigned gen = 0;  // counts recomputations: the spectra remember which bank they belong to
    } dc_bank;
    struct DcPlan {
        size_t M = 0;
        PlanHost H;
        c32 *d_tw = nullptr;
        size_t o1 = 0, o2 = 0, o3 = 0;
        bool f = false;
    } dc_plan;
    struct DcSpectra {
        c32 *d_H = nullptr;
        size_t M = 0;
        unsigned bank_gen = 0;
        int b0 = -1, b1 = -1;
        // tables of the Parseval band energies (k_dc_energy_pv), one block: [t1 512][t2 64][hpm nb 1024] c32, then g
        c32 *d_pv = nullptr;
        size_t pv_g_off = 0;  // in c32 units from d_pv
        int pv_gstride = 0;
    } dc_spectra;
    // the spectra of a slab's traces between the energies phase and the recombination phase of a group's
    // Deconvolution stage (thz_dc_slab_energies / thz_dc_slab_combine, deconv_api.cpp)
    struct DcSlab {
        c32 *d_spec = nullptr;
        size_t cap = 0;            // entries allocated
        size_t npix = 0, nk = 0;   // what it holds: npix rows of nk bins
        size_t M = 0;
    } dc_slab;
    void drop_dc_tables()
    {
        if (dc_plan.d_tw) (void)hipFree(dc_plan.d_tw);
        if (dc_spectra.d_H) (void)hipFree(dc_spectra.d_H);
        if (dc_spectra.d_pv) (void)hipFree(dc_spectra.d_pv);
        if (dc_slab.d_spec) (void)hipFree(dc_slab.d_spec);
        dc_plan = DcPlan{};
        dc_spectra = DcSpectra{};
        dc_slab = DcSlab{};
    }
    int timing = 0;  // 0 off, 1 immediate (host waits per call), 2 deferred (no host wait)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint64_t stage_ns[THZ_STAGE_COUNT] = {0};
    struct Rec {
        int stage;
        hipEvent_t a, b;
    };
    std::vector<Rec> recs;        // deferred records awaiting thz_timing_collect
    std::vector<hipEvent_t> pool;  // recycled events
};

// The Deconvolution stage in phases, for a group of GPUs (deconv_api.cpp; C++ linkage, not exported): energies and
// recombination run over a slab of pixels with every band, the iterations over the whole grid with a range of bands.
// nx, ny, dx, dy always describe the WHOLE (current) grid.  thz_dc_slab_energies returns THZ_SKIPPED when one of the
// reference's guards holds (the same on every rank): the stage then passes its input through.
int thz_dc_slab_energies(thz_ctx *ctx, const thz_psf *psf, const thz_deconv_cfg *cfg, size_t nx, size_t ny, float dx, float dy,
                         const float *d_in, size_t npix_local, float *d_energy /* [n_filters][npix_local] */);
int thz_dc_band_gains(thz_ctx *ctx, const thz_psf *psf, const thz_deconv_cfg *cfg /* band_begin, band_end */, size_t nx, size_t ny,
                      float dx, float dy, float *d_energy /* [bands][nx ny] */, float *d_gain /* [bands][nx ny] */,
                      volatile const int *abort_flag, float *progress);
int thz_dc_slab_combine(thz_ctx *ctx, const thz_psf *psf, const thz_deconv_cfg *cfg, size_t nx, size_t ny, float dx, float dy,
                        size_t npix_local, float *d_gain /* [n_filters][npix_local] */, float *d_out, float *d_img);
// host only: per band {iterations, iterations x tiles} (2 n_filters values), the same on every rank; THZ_SKIPPED under a guard
int thz_dc_band_costs(thz_ctx *ctx, const thz_psf *psf, const thz_deconv_cfg *cfg, size_t nx, size_t ny, float dx, float dy,
                      std::vector<double> *costs);

namespace thz_api {

inline int fail(thz_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                  \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(ctx, THZ_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

inline int use_device(thz_ctx *ctx)
{
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return THZ_OK;
}

inline int ensure_ws(thz_ctx *ctx, size_t bytes)
{
    if (ctx->ws_bytes >= bytes) return THZ_OK;
    if (ctx->ws) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipFree(ctx->ws));
        ctx->ws = nullptr;
        ctx->ws_bytes = 0;
    }
    HIP_TRY(ctx, hipMalloc(&ctx->ws, bytes));
    ctx->ws_bytes = bytes;
    return THZ_OK;
}

inline hipEvent_t pool_event(thz_ctx *ctx)
{
    if (!ctx->pool.empty()) {
        hipEvent_t e = ctx->pool.back();
        ctx->pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

// Brackets one stage call with events on the context's stream.
struct StageTimer {
    thz_ctx *ctx;
    int stage;
    hipEvent_t a = nullptr, b = nullptr;
    StageTimer(thz_ctx *c, int s) : ctx(c), stage(s)
    {
        if (ctx->timing == 1) (void)hipEventRecord(ctx->ev0, ctx->stream);
        if (ctx->timing == 2) {
            a = pool_event(ctx);
            b = pool_event(ctx);
            if (a) (void)hipEventRecord(a, ctx->stream);
        }
    }
    ~StageTimer()
    {
        if (ctx->timing == 1) {
            (void)hipEventRecord(ctx->ev1, ctx->stream);
            if (hipEventSynchronize(ctx->ev1) == hipSuccess) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess)
                    ctx->stage_ns[stage] = (uint64_t)((double)ms * 1e6);
            }
        } else if (ctx->timing == 2 && a && b) {
            (void)hipEventRecord(b, ctx->stream);
            ctx->recs.push_back({stage, a, b});
        }
    }
};

inline int check_launch(thz_ctx *ctx)
{
    HIP_TRY(ctx, hipGetLastError());
    return THZ_OK;
}

inline int need_plan(thz_ctx *ctx)
{
    if (!ctx) return THZ_ERR_INVALID;
    if (!ctx->have_plan) return fail(ctx, THZ_ERR_NOT_READY, "thz_set_time_axis has not been called");
    return use_device(ctx);
}

}  // namespace thz_api
using namespace thz_api;

// api.cpp: thz_pipeline_ex with the real multiplier's non-zero range [band_lo, band_hi) known (0, 0: unknown)
int pipeline_ex_band(thz_ctx *ctx, size_t npix, const thz_pipeline_io *io, size_t band_lo, size_t band_hi);
// api.cpp: order-free column sums over all rows (d_list null) or over the listed rows of d_arr
int pixel_sum_rows(thz_ctx *ctx, const float *d_arr, const uint32_t *d_list, size_t npix, size_t L, float *d_out);

