// kernels.hpp — plan descriptor and launcher prototypes shared by kernels.hip
// (device code) and api.cpp (C ABI).
#pragma once

#include "thz_device.hpp"

#include <stddef.h>

namespace thz {

constexpr int kNumCU = 256;                    // MI355X: 8 XCD x 32 CU
constexpr size_t kLdsBytesPerCU = 160 * 1024;  // gfx950 LDS per CU

enum : int { kModePow2 = 0, kModeBluestein = 1 };

// Passed by value to the transform kernels.
struct PlanDev {
    int nt;               // real trace length
    int nf;               // nt/2 + 1
    int mode;             // kModePow2: nt = 2N, half-length complex FFT + split
                          // kModeBluestein: chirp-z over a length-N complex FFT
    int log2n;            // complex transform length N = 1 << log2n
    int buf_entries;      // c32 entries per LDS ping-pong buffer
    int lds_per_wave;     // bytes (two buffers)
    int waves_per_block;  // 1..4
    const c32 *tw;        // exp(-2*pi*i*m/N), m in [0, N)
    const c32 *tw_split;  // pow2: exp(-2*pi*i*k/nt), k in [0, N/2]
    const c32 *chirp_conj;  // bluestein: exp(-i*pi*n^2/nt), n in [0, nt)
    const c32 *bfft;        // bluestein: FFT_N(chirp filter)/N, N entries
    // "F" family (register-resident three-pass transform, fft_f.hpp); used when
    // family == kFamilyF (nt = 1024 / 2048 / 4096)
    int family;
    const c32 *f_t1;   // [k1][m]    W_N^(m k1)
    const c32 *f_t2;   // [k2][j3]   W_(8 R2)^(j3 k2)
    const c32 *f_w2n;  // [k]        exp(-i*pi*k/N), k in [0, N)
    const float *ones; // nf floats of 1.0 (stand-in mask)
    // "P" family (mixed-radix three-pass transform of length nt = R1 R2 R3, fft_p.hpp); family == kFamilyP
    const c32 *p_t1;   // [k1][m]    W_nt^(m k1), nt entries
    const c32 *p_t2;   // [k2][j3]   W_(R2 R3)^(j3 k2)
    // "PH" kernels (fft_ph.hpp): an even length whose half N is a P plan — p_t1 / p_t2 are then those of length N, with
    // the split twiddles W_2N^k, k <= N / 2, behind p_t2's (even-padded) entries; the family stays the chirp-z one, whose
    // kernels serve what the PH kernels do not (a complex multiplier, thz_set_kernel_family)
    int half_n;        // N, or 0
    // lengths whose transform buffers do not fit the CU's LDS (not a power of two above 8191, powers of two above
    // 16384): the G kernels with their buffers in global scratch — 2 buf_entries per wave of a grid of big_waves waves
    c32 *big_scratch;  // or nullptr
    int big_waves;
};

enum : int { kFamilyG = 0, kFamilyF = 1, kFamilyFB = 2, kFamilyFB2 = 3, kFamilyFB4 = 4, kFamilyFB8 = 5, kFamilyP = 6 };  // FB / FB2: chirp-z over the F core (fft_fb.hpp)

// one band of the batched Richardson–Lucy solve (offsets in floats into one workspace)
struct RlBand {
    int h, w;           // image (nx, ny)
    int pr, pc;         // PSF rows (x) / cols (y)
    int pad_y, pad_x;   // pr/2, pc/2
    int H, W;           // padded image
    int n_iter;
    int mode;           // 0: <= 256-element kernel (correlation-indexed), 1: true "same" convolution
    unsigned blk0;      // first block of this band in the flattened grid (256 pixels per block)
    unsigned tblk0;     // first block of this band in the tiled grid (16 x 16 pixels per block)
    int tiles_w;        // tiles per row of the padded image
    int n_tiles;        // tiles of the padded image (a block holds one, or four for kernels of <= 256 taps)
    unsigned off_d, off_u, off_t, off_psf, off_mirror;
    // a wide kernel that is an outer product psf[m][n] = fx[m] fy[n] (every band PSF of the reference is one,
    // psf.rs:228-313): the two profiles, pr and pc floats — 0 / 0 when the kernel is only known as a 2-D array
    unsigned off_fx, off_fy;
    unsigned off_zero;  // a float that is 0.0f: what a halo position outside the image loads (no select, no mask to keep)
};
enum : int { kRlNarrow = 0, kRlWide = 1, kRlSeparable = 2 };  // kinds of tile lists (one kernel each)
// a tile: 16 x 16 pixels, but kRlSepTileRows x kRlSepTileCols for the separable kernels — their cost is the halo they
// fetch (a 47 x 57 kernel: 20 floats loaded per pixel of a 16 x 16 tile, 11 for 32 x 16, 6.7 for 32 x 32), not their
// arithmetic
constexpr int kRlTileRows = 16, kRlTileCols = 16, kRlSepTileRows = 32, kRlSepTileCols = 32;
inline int rl_tile_cols(int kind) { return kind == kRlSeparable ? kRlSepTileCols : kRlTileCols; }
inline int rl_tile_rows(int kind) { return kind == kRlSeparable ? kRlSepTileRows : kRlTileRows; }

void launch_dc_filter_spectra(hipStream_t st, const float *filters, int n_bands, int n_taps, const double *cs,
                              const double *sn, unsigned M, unsigned nk, c32 *H);
void launch_dc_fft(hipStream_t st, const PlanDev &P, size_t npix, int nt, const float *in, c32 *spec);
void launch_dc_energy(hipStream_t st, const PlanDev &P, size_t npix, int nt, int n_bands, int shift,
                      const c32 *spec, const c32 *H, float *energy);
// The same energies in Parseval form (round 3; k_dc_energy_pv in kernels.hip): full-convolution energy from |X|^2 and
// c_k M |H_b|^2, minus the energies of the first / last `shift` samples of the full convolution from ONE 512-point
// complex transform per band.  M in {1024, 2048, 4096}, odd tap count with 2 shift - 1 <= 512.
struct DcPvTables {
    const c32 *t1, *t2;  // FPlan1024 core tables (dc_pv_core_tables)
    const c32 *hpm;      // [n_bands][512][2]: (Hh + Ht) / 2, (Hh - Ht) / 2
    const float *g;      // [n_bands][gstride]
    int gstride;         // dc_pv_gstride(nk)
};
bool dc_energy_pv_supported(size_t M, int n_taps);
int dc_pv_gstride(int nk);
// H: [n_bands][nk] as launch_dc_filter_spectra leaves them; hht: [2 n_bands][512] the 512-point spectra (/ 512) of
// h_b[0 .. shift) and h_b[shift + 1 .. 2 shift]
void launch_dc_pv_tables(hipStream_t st, int n_bands, int nk, int gstride, size_t M, const c32 *H, const c32 *hht, float *g,
                         c32 *hpm);
void launch_dc_energy_pv(hipStream_t st, const DcPvTables &T, size_t npix, int nt, int n_bands, int shift, int nk,
                         const float *in, const c32 *spec, float *energy);
void launch_dc_combine(hipStream_t st, const PlanDev &P, size_t npix, int nt, int n_bands, int shift,
                       const c32 *spec, const c32 *H, const float *gain, float *out, float *img);
// padded lengths without an F core: y[p][k] = spec[p][k] sum_b gain[b gain_stride + p] H[b][k] as a kernel of its own
// (H slices in LDS), after which launch_dc_combine with n_bands = 0 only transforms y
bool dc_combine_has_f_core(const PlanDev &P, int nt, int shift);
bool dc_weight_spectra_supported(int n_bands);
void launch_dc_weight_spectra(hipStream_t st, size_t npix, size_t gain_stride, int nk, int n_bands, const c32 *spec,
                              const c32 *H, const float *gain, c32 *y);
void launch_rl_init(hipStream_t st, const RlBand *d_bands, int n_bands, unsigned total_blocks,
                    size_t npix, const float *energy, float *ws);
// iteration = (it_base ? *it_base : 0) + iteration: a captured batch is replayed with a new base
void launch_rl_step(hipStream_t st, const RlBand *d_bands, int n_bands, unsigned total_blocks,
                    const int *it_base, int iteration, int step, float *ws);
// LDS-tiled form of the same step: total_tiles blocks, lds_bytes = rl_tile_lds_bytes of the largest band
size_t rl_tile_lds_bytes(int pr, int pc, bool separable = false);
unsigned rl_tile_block_count(int pr, int pc, unsigned n_tiles);  // blocks of the tiled grid a band's tiles take
void prepare_rl_step_tiled(int kind, size_t lds_bytes);  // raises the kernel's dynamic-LDS limit (not capturable)
// per tile of the tiled grid: its band's record by value — a block needs one (scalar) load to know
// whether its band still iterates and everything else about it, not a chain of two
struct alignas(16) RlTileRef {
    RlBand band;
    int pad[24 - sizeof(RlBand) / sizeof(int)];  // 96 bytes: the kernels fetch a record as sixteen + eight dwords
};
static_assert(sizeof(RlBand) == 22 * sizeof(int) && sizeof(RlTileRef) == 96, "rl_block_band reads the record as 24 dwords");
void launch_rl_step_tiled(hipStream_t st, int kind, const RlTileRef *d_tiles, unsigned total_tiles, size_t lds_bytes,
                          const int *it_base, int iteration, int step, float *ws);
void launch_dc_gain(hipStream_t st, const RlBand *d_bands, int n_bands, size_t npix,
                    const float *energy, const float *ws, float *gain);

void launch_fft_fwd(hipStream_t st, const PlanDev &P, size_t npix, const float *in,
                    const float *wa, const float *wb, float *data_out, c32 *fft_out,
                    float *amp_out, float *ph_out, const float *mask, const c32 *cmask = nullptr);
void launch_fft_inv(hipStream_t st, const PlanDev &P, size_t npix, const c32 *fft_in,
                    const float *win, float *out, float *img);
// sum_partial: pipeline_sum_rows(P, npix, cmask != nullptr) rows of 2 nf floats, or null — every block's sums of the
// stored amplitudes | unwrapped phases of its traces (the F kernels' kCfgSums, fft_f.hpp); a second small launch
// (launch_sum_axis0 over the rows) makes the pixel sums of them
// band_lo4 / band_n (multiples of 4; 0, 0 = unknown): the bins outside [band_lo4, band_lo4 + band_n) are zero in `mask` —
// lets the nt = 4096 chain with a complex multiplier and the sums stage a band-limited table (fft_f.hpp, kCfgBand)
size_t pipeline_sum_rows(const PlanDev &P, size_t npix, bool cmask, int band_lo4 = 0, int band_n = 0);
void launch_pipeline(hipStream_t st, const PlanDev &P, size_t npix, const float *raw,
                     const float *pre_win, const float *mask, const float *post_win, c32 *fft_out,
                     float *amp_out, float *ph_out, float *data_out, float *img, const c32 *cmask = nullptr,
                     float *sum_partial = nullptr, int band_lo4 = 0, int band_n = 0);
void launch_fd_mask(hipStream_t st, size_t npix, int nf, c32 *fft, float *amp, const float *mask);
void launch_fd_cmask(hipStream_t st, size_t npix, int nf, int nt, c32 *fft, float *amp,
                     const c32 *cmask);
void launch_td_window(hipStream_t st, size_t npix, int nt, const float *in, const float *win,
                      float *out);
void launch_scale_vec(hipStream_t st, const float *in, float f, size_t n, float *out);          // out = in * f
void launch_add_vec(hipStream_t st, float *dst, const float *src, size_t n);                    // dst += src
void launch_add_u64(hipStream_t st, unsigned long long *dst, const unsigned long long *src, size_t n);
void launch_intensity(hipStream_t st, size_t npix, int nt, float *data, float *img,
                      int subtract_bias);
// carry (or null; may alias out): the sequential sum continues from it
void launch_sum_axis0(hipStream_t st, const float *arr, size_t n0, size_t inner, float div,
                      float *out, const float *carry = nullptr);
// m rows of a block of s rows: out[(ny / s) x L] = (carry +) their column-block sums, divided by div when > 0
void launch_scale_rows_partial(hipStream_t st, const float *arr, size_t m, size_t ny, size_t L, size_t s, const float *carry, float div,
                               float *out);
void launch_sum_rows_f64(hipStream_t st, const float *arr, size_t n0, size_t inner, float *out);  // column sums of a few rows, adds in double
// list (or null): add arr's rows list[0 .. nrows) instead of rows 0 .. nrows - 1 (a region of interest's pixels)
size_t launch_colsum_partial(hipStream_t st, const float *arr, size_t nrows, size_t L,
                             float *partial, size_t max_groups, const uint32_t *list = nullptr);
void launch_roi_mask(hipStream_t st, const uint64_t *d_poly, int n, uint64_t x_min, uint64_t x_max,
                     uint64_t y_min, uint64_t y_max, uint64_t x_size, uint64_t y_size,
                     uint8_t *d_mask);
void launch_gather_sum(hipStream_t st, const float *arr, size_t len, const uint32_t *d_list,
                       uint32_t count, float div, float *out);
// launch_gather_sum over arr * w1 * w2 * w3 (each factor optional, one f32 multiply each, in this order)
void launch_gather_sum_w(hipStream_t st, const float *arr, size_t len, const uint32_t *d_list, uint32_t count, float div,
                         const float *w1, const float *w2, const float *w3, float *out);
void launch_div_vec(hipStream_t st, const float *in, const float *w, float d, size_t n, float *out);  // out = (in [* w]) / d, IEEE division
void launch_scale3d(hipStream_t st, const float *arr, size_t nx, size_t ny, size_t L, size_t s,
                    float *out);

void launch_tilt(hipStream_t st, size_t npix, int nt_in, int nt_out, const float *in,
                 const float *taper, const int *insert_index, float *out);
// ---- K15 voxel envelope (voxel.hip)
constexpr int kVoxPad = 12;                // register-window path: radius <= kVoxPad
constexpr int kVoxTaps = 2 * kVoxPad + 1;  // zero-padded tap vector
constexpr int kVoxMaxNt = 8192;
constexpr int kSelBins = 2048;
struct VoxelTaps {
    float c[kVoxTaps];
};
struct VoxelGeom {
    float spacing_w, spacing_h, spacing_d, half_w, half_h, half_d, scale, threshold;
    size_t x0;  // first x-row of this tile in the whole grid
};
bool launch_voxel_opacity(hipStream_t st, size_t npix, int nt, const float *data, const VoxelTaps &taps,
                          const float *wide_taps, int radius, float contrast, float opacity_threshold,
                          float *out);
void launch_select_hist(hipStream_t st, const float *vals, size_t n, int level, uint32_t prefix,
                        unsigned long long *hist);
void launch_voxel_count(hipStream_t st, size_t npix, int nt, const float *opacity, float threshold,
                        uint32_t *counts);
void launch_scan_counts(hipStream_t st, const uint32_t *counts, size_t n, unsigned long long *tile_ws,
                        unsigned long long *offsets, unsigned long long *total);
void launch_voxel_emit(hipStream_t st, size_t npix, int nt, size_t gh, const float *opacity,
                       const uint32_t *counts, const unsigned long long *offsets, const VoxelGeom &g, float *out,
                       unsigned long long capacity);

void launch_traffic_probe(hipStream_t st, size_t npix, int nt, const float *in, float *fft, float *amp, float *ph,
                          float *out);

void launch_synth(hipStream_t st, float *out, size_t ntraces, int nt, uint64_t first_trace,
                  const float *time, uint32_t seed, int subtract_bias);

}  // namespace thz
