// session.hpp — the resident-cube session behind thz_session* (session_api.cpp) and the pieces of its
// recompute that the multi-GPU group (group_api.cpp) drives slab by slab.
#pragma once
#include "ctx.hpp"

#include <vector>

struct thz_session {
    thz_ctx *ctx = nullptr;
    size_t nx = 0, ny = 0, nt = 0, nf = 0;
    float dx = 1.0f, dy = 1.0f;
    // grid of everything behind the scaling stage (math_tools.rs:242-310): nx / s, ny / s, dx * s, dy * s
    size_t nx_cur = 0, ny_cur = 0, scale = 1, out_pix = 0;
    float dx_cur = 1.0f, dy_cur = 1.0f;
    float *d_scaled = nullptr;                 // block-averaged raw cube when scale > 1
    size_t scaled_floats = 0;
    std::vector<float> time, time_out;
    size_t nt_out = 0, nf_out = 0;
    float *d_raw = nullptr, *d_fft = nullptr, *d_amp = nullptr, *d_ph = nullptr, *d_data = nullptr,
          *d_img = nullptr, *d_avg = nullptr;  // d_avg: [2 nf | nf | nf]
    float *d_vec = nullptr;                    // pre | mask | post multipliers (+ tilt scratch)
    float *h_vec = nullptr;                    // pinned host image of d_vec: the multipliers go up in one asynchronous copy
    size_t vec_floats = 0;
    float *d_tilt = nullptr;                   // extended cube when tilt != 0 (kept while its size stays the same)
    size_t tilt_floats = 0, ins_count = 0;
    std::vector<float> fd_real, fd_cmask;      // further Frequency-domain plugins: K14 real (nf), K13 complex (2 nf)
    thz_chain_cfg last_cfg{};                  // configuration of the last full recompute (decides whether a
    bool have_last_cfg = false;                // start position >= 6 may reuse the resident spectrum)
    float *d_deconv = nullptr, *d_deconv_img = nullptr;  // output of the Deconvolution stage (thz_session_deconvolve)
    size_t deconv_floats = 0;
    bool deconv_current = false;               // ... and whether it is the chain's final output right now
    float *d_opacity = nullptr;                // voxel opacities of the final cube (thz_session_voxels)
    size_t opacity_floats = 0;
    int32_t *d_ins = nullptr;
    float *d_rawsum = nullptr;   // (nt) sum over the pixels of the raw (bias-subtracted) traces, taken at upload
    float *d_msum = nullptr;     // [Σ source trace: nt_out | Σ amplitudes: nf | Σ phases: nf] of the last recompute, undivided
    size_t msum_floats = 0;
    bool msum_fast = false;      // the last recompute left its undivided sums in d_msum (want_means == 1)
    bool have_means = false;
    bool have_outputs = false;   // a recompute has run
    const float *d_src = nullptr;  // what the fft stage read: d_raw, d_scaled or d_tilt (extended axis)
};


// first half of a recompute: everything up to and including the fused launch, enqueued on the context's
// stream (tail_only: chain positions >= 6 were served from the resident spectrum)
int session_enqueue(thz_session *s, const thz_chain_cfg *cfg, int start_stage, bool *tail_only);
// second half: pixel means from the sums in d_msum, which cover total_pix pixels (the slab's own, or all
// slabs' after the group's all-reduce)
int session_means(thz_session *s, const thz_chain_cfg *cfg, size_t total_pix);
