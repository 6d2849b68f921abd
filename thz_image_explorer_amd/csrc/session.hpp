// session.hpp — the resident-cube session behind thz_session* (session_api.cpp) and the pieces of its
// recompute that the multi-GPU group (group_api.cpp) drives slab by slab.
#pragma once
#include "ctx.hpp"

#include <vector>

// One region of interest of a session (thz_session_set_rois) and its pixel list for the grid of the last recompute
struct SessionRoi {
    std::vector<uint64_t> poly;   // (x, y) pairs as the reference keeps them: pixels of the raw grid
    uint32_t *d_list = nullptr;   // THIS slab's pixels inside, in the reference's visiting order (y outer, x inner;
    size_t list_cap = 0;          //   mask position (x, y) samples pixel [shape0 - y - 1, x], math_tools.rs:640-651)
    uint32_t count = 0;           // ... how many
    uint32_t total = 0;           // pixels of the WHOLE grid inside (= count unless the session is a group's slab)
    size_t for_rows = 0, for_cols = 0, for_scale = 0, for_x0 = 0, for_grid_rows = 0;  // what the list was built for
};

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
    // ---- regions of interest (session_roi.cpp)
    std::vector<SessionRoi> rois;
    float *d_roi_sum = nullptr;    // [amplitudes R x nf | phases R x nf | stage-input traces R x nt | final traces R x nt]:
    float *d_roi = nullptr;        //   this session's (after a group's all-reduce: the grid's) sums, and the means
    size_t roi_floats = 0;
    float *d_wsep = nullptr;       // the three time multipliers in front of the transform, one by one (reference-order
    size_t wsep_floats = 0;        //   roi_data): [tilt taper | Time Band Pass | fft window], nt_out each
    bool wsep_on[3] = {false, false, false};
    std::vector<float> roi_polar;  // avg_in_fourier_space: per region the polar inverse transform (nt_out each), host
    std::vector<char> roi_polar_ok;  //   0: realfft would have refused the spectrum (the reference falls back to the traces)
    std::vector<float> avg_data;   // avg_in_fourier_space: the ifft stage's avg_data (nt_out), host
    bool have_rois = false;        // d_roi holds the means of the last recompute
    // The regions' sums of the SOURCE traces (parallel sums) do not depend on the sliders: they are kept for as long as
    // the source cube, the grid and the regions stay what they were (src_gen counts uploads, scale / tilt changes and
    // list rebuilds; roi_src_gen is the generation the kept sums belong to, 0 = none)
    unsigned long src_gen = 1, roi_src_gen = 0;
    bool roi_src_fresh = false;    // the last session_roi_sums renewed the source block (a group all-reduces it then)
    int last_sf = -1, last_tilt_active = -1;
    double last_tilt_x = 0.0, last_tilt_y = 0.0;
    // Placement in a group's grid (group_api.cpp).  raw_grid_rows == 0: the session is the whole grid.  Otherwise it
    // holds rows raw_grid_x0 .. + nx of the raw_grid_rows rows of the RAW grid (slab `slab_rank` of `slab_world`, all
    // slabs cut by thz_host_slab), and session_enqueue derives grid_x0 / grid_rows: the same for the CURRENT grid —
    // behind a scaling stage the block grid, whose rows belong to the slab that holds a block's LAST raw row.
    size_t raw_grid_x0 = 0, raw_grid_rows = 0;
    int slab_rank = 0, slab_world = 1;
    size_t grid_x0 = 0, grid_rows = 0;
    // scaling over slab edges: the undivided partial sums of the block this slab only holds the first rows of (handed
    // to the next slab by the group), and the previous slab's for the block this one completes; (ny / s) x nt each
    float *d_carry_out = nullptr, *d_carry_in = nullptr;
    size_t carry_floats = 0;
    bool carry_out_valid = false;
    bool msum_passes = false;    // a group's slab, tilted: d_msum = [unused nt | sum fft 2 nf | sum amplitudes nf | sum phases nf]
};


// first half of a recompute: everything up to and including the fused launch, enqueued on the context's
// stream (tail_only: chain positions >= 6 were served from the resident spectrum)
int session_enqueue(thz_session *s, const thz_chain_cfg *cfg, int start_stage, bool *tail_only);
// Scaling over slab edges (group_api.cpp; math_tools.rs:273-301 adds a block's s x s inputs row by row): which rows
// of the block grid slab `rank` of `world` owns when the raw grid's nx rows are cut by thz_host_slab.
struct SlabScale {
    size_t head = 0;       // leading raw rows that complete the block the previous slab started (0: none)
    bool head_valid = false;  // ... and that block lies inside the block grid
    size_t full = 0;       // blocks that lie wholly inside the slab
    size_t tail = 0;       // trailing raw rows that start a block the next slab completes (0: none, or beyond the block grid)
    size_t rows = 0;       // rows of the block grid this slab owns = head_valid + full
    size_t x0 = 0;         // ... and where they start in the block grid
    bool ok = true;        // false: a slab shorter than the scale factor (a block would span three slabs)
};
SlabScale slab_scale(size_t nx_total, int world, int rank, size_t sf);
// before session_enqueue of a scaled group recompute: the partial sums of this slab's trailing block into d_carry_out
int session_scale_tail(thz_session *s, const thz_chain_cfg *cfg);
// second half: pixel means from the sums in d_msum, which cover total_pix pixels (the slab's own, or all
// slabs' after the group's all-reduce)
int session_means(thz_session *s, const thz_chain_cfg *cfg, size_t total_pix);
// Regions of interest, in two halves like the means: session_roi_sums enqueues the masked sums of this session's
// pixels into d_roi_sum (*data_only: the chain's tail or the Deconvolution stage changed the final traces only —
// cleared when the buffers are new and everything is summed after all);
// a group all-reduces d_roi (thz_session_roi_floats floats) in between; session_roi_finish divides by the
// regions' pixel counts over the whole grid and, with avg_in_fourier_space, takes the polar inverse transforms.
int session_roi_sums(thz_session *s, const thz_chain_cfg *cfg, bool *data_only);
int session_roi_finish(thz_session *s, const thz_chain_cfg *cfg, bool data_only);
size_t session_roi_floats(const thz_session *s);
void session_roi_free(thz_session *s);
// the ifft stage's avg_data (avg_in_fourier_space; needs the means): host-side, after session_means
int session_avg_data(thz_session *s, const thz_chain_cfg *cfg);
