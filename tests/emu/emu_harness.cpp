// emu_harness.cpp — C entry points over the product's kernel launchers,
// built with -DTHZ_EMU (host threads instead of a GPU).  TEST INFRASTRUCTURE
// ONLY: lets tests/test_emu_kernels.py check the kernels' index arithmetic in a
// container without a GPU.  "Device" pointers are host pointers here.
#include "plan_host.hpp"

using namespace thz;

namespace thz { extern int g_f_bar_override; extern int g_p_pairs_override; extern int g_grid_cap_override; }

static int g_allow_f = 1, g_allow_p = 1;
static std::vector<float> g_ones;
static std::vector<c32> g_big;  // "global scratch" of a plan whose buffers do not fit LDS
static PlanDev make_plan(PlanHost &H)
{
    g_ones.assign((size_t)H.nf, 1.0f);
    if (H.big) g_big.assign((size_t)H.big_waves * (size_t)H.lds_per_wave / sizeof(c32), c32{0.0f, 0.0f});
    return plan_dev(H, H.tw.data(), H.tw_split.data(), H.chirp_conj.data(), H.bfft.data(),
                    H.f_t1.empty() ? nullptr : H.f_t1.data(), H.f_t2.empty() ? nullptr : H.f_t2.data(),
                    H.f_w2n.empty() ? nullptr : H.f_w2n.data(), g_ones.data(), H.p_t1.empty() ? nullptr : H.p_t1.data(),
                    H.p_t2.empty() ? nullptr : H.p_t2.data(), H.big ? g_big.data() : nullptr);
}

extern "C" {

void emu_allow_f(int on) { g_allow_f = on; }
void emu_allow_p(int on) { g_allow_p = on; }
int emu_family(int nt)
{
    PlanHost H;
    if (!build_plan((size_t)nt, H, g_allow_f != 0, g_allow_p != 0)) return -2;
    return H.family;
}

int emu_half_n(int nt)
{
    PlanHost H;
    if (!build_plan((size_t)nt, H, g_allow_f != 0, g_allow_p != 0)) return -2;
    return H.half_n;
}

int emu_fft_fwd(int nt, size_t npix, const float *in, const float *wa, const float *wb,
                float *data_out, float *fft, float *amp, float *ph, const float *mask)
{
    PlanHost H;
    if (!build_plan((size_t)nt, H, g_allow_f != 0, g_allow_p != 0)) return -2;
    PlanDev D = make_plan(H);
    launch_fft_fwd(nullptr, D, npix, in, wa, wb, data_out, (c32 *)fft, amp, ph, mask);
    return 0;
}

int emu_fft_inv(int nt, size_t npix, const float *fft, const float *win, float *out, float *img)
{
    PlanHost H;
    if (!build_plan((size_t)nt, H, g_allow_f != 0, g_allow_p != 0)) return -2;
    PlanDev D = make_plan(H);
    launch_fft_inv(nullptr, D, npix, (const c32 *)fft, win, out, img);
    return 0;
}

int emu_pipeline(int nt, size_t npix, const float *raw, const float *pre, const float *mask,
                 const float *post, float *fft, float *amp, float *ph, float *out, float *img)
{
    PlanHost H;
    if (!build_plan((size_t)nt, H, g_allow_f != 0, g_allow_p != 0)) return -2;
    if (H.mode != kModePow2 && H.family < kFamilyFB && !H.big) return -2;
    PlanDev D = make_plan(H);
    launch_pipeline(nullptr, D, npix, raw, pre, mask, post, (c32 *)fft, amp, ph, out, img);
    return 0;
}

void emu_set_f_bar(int mode) { thz::g_f_bar_override = mode; }
void emu_set_p_pairs(int q) { thz::g_p_pairs_override = q; }
void emu_set_grid_cap(int blocks) { thz::g_grid_cap_override = blocks; }

// fused chain with a complex multiplier (F and P families); FArgs::bar via emu_set_f_bar
int emu_pipeline_ex(int nt, size_t npix, const float *raw, const float *pre, const float *mask, const float *cmask,
                    const float *post, float *fft, float *amp, float *ph, float *out, float *img)
{
    PlanHost H;
    if (!build_plan((size_t)nt, H, g_allow_f != 0, g_allow_p != 0)) return -2;
    if (H.family != kFamilyF && H.family != kFamilyP) return -2;
    PlanDev D = make_plan(H);
    launch_pipeline(nullptr, D, npix, raw, pre, mask, post, (c32 *)fft, amp, ph, out, img, (const c32 *)cmask);
    return 0;
}

// fused chain with the pixel sums taken inside the launch (F family: kCfgSums / FSums; P family: PSums): sums[0, nf) amplitudes, [nf, 2 nf)
// unwrapped phases, as thz_pipeline_ex does it (block rows + launch_sum_axis0).  Returns the number of block rows.
int emu_pipeline_sums(int nt, size_t npix, const float *raw, const float *pre, const float *mask, const float *cmask,
                      const float *post, float *fft, float *amp, float *ph, float *out, float *img, float *sums)
{
    PlanHost H;
    if (!build_plan((size_t)nt, H, g_allow_f != 0, g_allow_p != 0)) return -2;
    if (H.family != kFamilyF && H.family != kFamilyP) return -2;
    PlanDev D = make_plan(H);
    const size_t rows = pipeline_sum_rows(D, npix, cmask != nullptr);
    if (rows == 0) return -3;
    std::vector<float> partial(rows * 2 * (size_t)D.nf, -777.0f);  // every entry must be written by the kernel
    launch_pipeline(nullptr, D, npix, raw, pre, mask, post, (c32 *)fft, amp, ph, out, img, (const c32 *)cmask, partial.data());
    launch_sum_rows_f64(nullptr, partial.data(), rows, 2 * (size_t)D.nf, sums);
    return (int)rows;
}

// ... with the caller's word on where the real mask is not zero ([band_lo, band_hi), api.cpp pipeline_ex_band): at nt = 4096 with a
// complex multiplier the kernel then stages a band-limited table (kCfgBand)
int emu_pipeline_sums_band(int nt, size_t npix, const float *raw, const float *pre, const float *mask, const float *cmask,
                           const float *post, float *fft, float *amp, float *ph, float *out, float *img, float *sums, int band_lo,
                           int band_hi)
{
    PlanHost H;
    if (!build_plan((size_t)nt, H, g_allow_f != 0, g_allow_p != 0)) return -2;
    if (H.family != kFamilyF) return -2;
    PlanDev D = make_plan(H);
    const int lo4 = band_lo & ~3, n = ((band_hi + 3) & ~3) - lo4;
    const size_t rows = pipeline_sum_rows(D, npix, cmask != nullptr, lo4, n);
    if (rows == 0) return -3;
    std::vector<float> partial(rows * 2 * (size_t)D.nf, -777.0f);
    launch_pipeline(nullptr, D, npix, raw, pre, mask, post, (c32 *)fft, amp, ph, out, img, (const c32 *)cmask, partial.data(), lo4, n);
    launch_sum_rows_f64(nullptr, partial.data(), rows, 2 * (size_t)D.nf, sums);
    return (int)rows;
}

int emu_intensity(size_t npix, int nt, float *data, float *img, int subtract_bias)
{
    launch_intensity(nullptr, npix, nt, data, img, subtract_bias);
    return 0;
}

int emu_roi_mask(const uint64_t *poly, int n, uint64_t x_min, uint64_t x_max, uint64_t y_min,
                 uint64_t y_max, uint64_t x_size, uint64_t y_size, uint8_t *mask)
{
    launch_roi_mask(nullptr, poly, n, x_min, x_max, y_min, y_max, x_size, y_size, mask);
    return 0;
}
}

extern "C" void emu_fast_atan2(const float *y, const float *x, int n, float *out)
{
    for (int i = 0; i < n; ++i) out[i] = thz::fast_atan2f(y[i], x[i]);
}

extern "C" void emu_div_const(const float *x, int n, float d, float *out)
{
    const thz::DivConst div(d);
    for (int i = 0; i < n; ++i) out[i] = div(x[i]);
}

// ---- K15 voxel envelope (voxel.hip)
extern "C" {

int emu_voxel_opacity(size_t npix, int nt, const float *data, const float *kernel, int radius, float contrast,
                      float opacity_threshold, float *out)
{
    VoxelTaps taps;
    for (int i = 0; i < kVoxTaps; ++i) taps.c[i] = 0.0f;
    if (radius <= kVoxPad)
        for (int i = 0; i <= 2 * radius; ++i) taps.c[i + (kVoxPad - radius)] = kernel[i];
    return launch_voxel_opacity(nullptr, npix, nt, data, taps, kernel, radius, contrast, opacity_threshold, out) ? 0 : -2;
}

int emu_select_hist(const float *vals, size_t n, int level, uint32_t prefix, unsigned long long *hist)
{
    launch_select_hist(nullptr, vals, n, level, prefix, hist);
    return 0;
}

int emu_voxel_instances(size_t npix, int nt, size_t gh, const float *opacity, const float *geom8, size_t x0,
                        float *out, unsigned long long capacity, unsigned long long *total)
{
    std::vector<uint32_t> counts(npix);
    std::vector<unsigned long long> offsets(npix), tiles((npix + 2047) / 2048 + 1);
    VoxelGeom g{geom8[0], geom8[1], geom8[2], geom8[3], geom8[4], geom8[5], geom8[6], geom8[7], x0};
    launch_voxel_count(nullptr, npix, nt, opacity, g.threshold, counts.data());
    launch_scan_counts(nullptr, counts.data(), npix, tiles.data(), offsets.data(), total);
    launch_voxel_emit(nullptr, npix, nt, gh, opacity, counts.data(), offsets.data(), g, out, capacity);
    return 0;
}
}

// One Richardson-Lucy iteration (both steps) of ONE band on padded images d, u (H x W, H = h + 2 (pr/2),
// W = w + 2 (pc/2)), through the untiled kernel (tiled = 0) or the LDS-tiled one: t_out = d / (u (*) psf + eps),
// u_out = u * (t (*) mirror).  Workspace laid out as thz_deconvolve does it.
// tiled = 2: psf is the outer product fx (pr) x fy (pc) and the step runs as two 1-D passes (k_rl_step_sep)
extern "C" int emu_rl_iteration_sep(int h, int w, int pr, int pc, int mode, const float *psf, const float *fx,
                                    const float *fy, const float *d, const float *u, int tiled, float *t_out,
                                    float *u_out)
{
    RlBand B{};
    B.h = h; B.w = w; B.pr = pr; B.pc = pc; B.pad_y = pr / 2; B.pad_x = pc / 2;
    B.H = h + 2 * B.pad_y; B.W = w + 2 * B.pad_x;
    B.n_iter = 1; B.mode = mode; B.blk0 = 0; B.tblk0 = 0; B.tiles_w = (B.W + 15) / 16;
    const size_t img = (size_t)B.H * B.W, taps = (size_t)pr * pc;
    B.off_d = 0; B.off_u = (unsigned)img; B.off_t = (unsigned)(2 * img);
    B.off_psf = (unsigned)(3 * img); B.off_mirror = (unsigned)(3 * img + taps);
    B.off_zero = (unsigned)(3 * img + 2 * taps);  // the 32 zeros behind the taps
    std::vector<float> ws(3 * img + 2 * taps + 32 + (size_t)pr + (size_t)pc, 0.0f);
    if (fx && fy) {
        B.off_fx = (unsigned)(3 * img + 2 * taps + 32);
        B.off_fy = B.off_fx + (unsigned)pr;
        std::memcpy(ws.data() + B.off_fx, fx, (size_t)pr * sizeof(float));
        std::memcpy(ws.data() + B.off_fy, fy, (size_t)pc * sizeof(float));
    } else if (tiled == 2) return -2;
    std::memcpy(ws.data() + B.off_d, d, img * sizeof(float));
    std::memcpy(ws.data() + B.off_u, u, img * sizeof(float));
    for (size_t i = 0; i < taps; ++i) {
        ws[B.off_psf + i] = psf[i];
        ws[B.off_mirror + i] = psf[taps - 1 - i];
    }
    const unsigned blocks = (unsigned)((img + 255) / 256);
    const int tile_rows = rl_tile_rows(tiled == 2 ? kRlSeparable : kRlNarrow);
    B.tiles_w = (B.W + rl_tile_cols(tiled == 2 ? kRlSeparable : kRlNarrow) - 1) / rl_tile_cols(tiled == 2 ? kRlSeparable : kRlNarrow);
    B.n_tiles = B.tiles_w * ((B.H + tile_rows - 1) / tile_rows);
    const unsigned tiles_n = rl_tile_block_count(pr, pc, (unsigned)B.n_tiles);  // blocks of the tiled grid
    std::vector<RlTileRef> tiles(tiles_n, RlTileRef{B});
    for (int step = 0; step < 2; ++step) {
        if (tiled == 2) launch_rl_step_tiled(nullptr, kRlSeparable, tiles.data(), tiles_n, rl_tile_lds_bytes(pr, pc, true), nullptr, 0, step, ws.data());
        else if (tiled) launch_rl_step_tiled(nullptr, mode != 0 ? kRlWide : kRlNarrow, tiles.data(), tiles_n, rl_tile_lds_bytes(pr, pc), nullptr, 0, step, ws.data());
        else launch_rl_step(nullptr, &B, 1, blocks, nullptr, 0, step, ws.data());
    }
    std::memcpy(t_out, ws.data() + B.off_t, img * sizeof(float));
    std::memcpy(u_out, ws.data() + B.off_u, img * sizeof(float));
    return 0;
}

extern "C" int emu_rl_iteration(int h, int w, int pr, int pc, int mode, const float *psf, const float *d,
                                const float *u, int tiled, float *t_out, float *u_out)
{
    return emu_rl_iteration_sep(h, w, pr, pc, mode, psf, nullptr, nullptr, d, u, tiled, t_out, u_out);
}

// The deconvolution's transform kernels on the padded length M (a power of two): forward transform of the
// zero-padded traces, band energies over the "same" slice and the gain-weighted recombination — with the
// register-resident F core (use_f != 0, M = 1024 / 2048 / 4096) or the generic LDS transform.
extern "C" int emu_dc_chain(int M, int nt, size_t npix, int n_bands, int shift, const float *in, const float *H,
                            const float *gain, int use_f, float *energy, float *out, float *img)
{
    PlanHost P;
    if (!build_plan((size_t)M, P, use_f != 0)) return -2;
    PlanDev D = plan_dev(P, P.tw.data(), P.tw_split.data(), nullptr, nullptr);
    if (use_f) {
        if (P.f_t1.empty()) return -3;
        D.f_t1 = P.f_t1.data();
        D.f_t2 = P.f_t2.data();
        D.f_w2n = P.f_w2n.data();
    }
    const size_t nk = (size_t)M / 2 + 1;
    std::vector<c32> spec(npix * nk);
    launch_dc_fft(nullptr, D, npix, nt, in, spec.data());
    launch_dc_energy(nullptr, D, npix, nt, n_bands, shift, spec.data(), (const c32 *)H, energy);
    if (use_f == 0 && dc_weight_spectra_supported(n_bands)) {
        // the recombination of padded lengths without an F core as deconv_api.cpp runs it: the multiplier sum as a kernel of
        // its own, then the generic transform alone; checked against the one-kernel form
        std::vector<c32> y(npix * nk);
        std::vector<float> out1(npix * (size_t)nt), img1(npix);
        launch_dc_combine(nullptr, D, npix, nt, n_bands, shift, spec.data(), (const c32 *)H, gain, out1.data(), img1.data());
        launch_dc_weight_spectra(nullptr, npix, npix, (int)nk, n_bands, spec.data(), (const c32 *)H, gain, y.data());
        launch_dc_combine(nullptr, D, npix, nt, 0, shift, y.data(), nullptr, nullptr, out, img);
        double worst = 0.0, top = 0.0;
        for (size_t i = 0; i < out1.size(); ++i) {
            worst = std::max(worst, (double)std::fabs(out1[i] - out[i]));
            top = std::max(top, (double)std::fabs(out1[i]));
        }
        if (worst > 2e-6 * top) return -5;
        return 0;
    }
    launch_dc_combine(nullptr, D, npix, nt, n_bands, shift, spec.data(), (const c32 *)H, gain, out, img);
    return 0;
}

// ---- the bandwidth-shaped helper kernels (multiplier, column sums, pixel-list sums, block means, tilt copy)
extern "C" {

int emu_td_window(size_t npix, int nt, const float *in, const float *win, float *out)
{
    launch_td_window(nullptr, npix, nt, in, win, out);
    return 0;
}

// thz_pixel_sum's launch sequence (api.cpp): column sums of an (nrows, L) array
int emu_pixel_sum(size_t nrows, size_t L, const float *arr, float *out)
{
    if (nrows < 64) {
        launch_sum_axis0(nullptr, arr, nrows, L, 0.0f, out);
        return 0;
    }
    const size_t max_groups = 2048, mid_groups = 32;
    std::vector<float> part((max_groups + mid_groups) * L);
    float *part2 = part.data() + max_groups * L;
    size_t groups = launch_colsum_partial(nullptr, arr, nrows, L, part.data(), max_groups);
    if (groups == 0) {
        launch_sum_axis0(nullptr, arr, nrows, L, 0.0f, out);
        return 1;
    }
    const float *src = part.data();
    if (groups > 4 * mid_groups) {
        groups = launch_colsum_partial(nullptr, part.data(), groups, L, part2, mid_groups);
        src = part2;
    }
    launch_sum_axis0(nullptr, src, groups, L, 0.0f, out);
    return 0;
}

int emu_gather_sum(const float *arr, size_t len, const uint32_t *list, uint32_t count, float div, float *out)
{
    launch_gather_sum(nullptr, arr, len, list, count, div, out);
    return 0;
}

int emu_scale3d(const float *arr, size_t nx, size_t ny, size_t L, size_t s, float *out)
{
    launch_scale3d(nullptr, arr, nx, ny, L, s, out);
    return 0;
}

int emu_tilt(size_t npix, int nt_in, int nt_out, const float *in, const float *taper, const int *ins, float *out)
{
    launch_tilt(nullptr, npix, nt_in, nt_out, in, taper, ins, out);
    return 0;
}
}

extern "C" int emu_dc_filter_spectra(const float *filters, int n_bands, int n_taps, int M, float *H)
{
    std::vector<double> cs((size_t)M), sn((size_t)M);
    for (int m = 0; m < M; ++m) {
        const double a = -2.0 * 3.14159265358979323846 * (double)m / (double)M;
        cs[(size_t)m] = std::cos(a);
        sn[(size_t)m] = std::sin(a);
    }
    launch_dc_filter_spectra(nullptr, filters, n_bands, n_taps, cs.data(), sn.data(), (unsigned)M, (unsigned)(M / 2 + 1),
                             (c32 *)H);
    return 0;
}

// The band energies in Parseval form (k_dc_energy_pv) from the filters themselves: spectra at M, the 512-point
// spectra of the two filter halves, the G / Hp / Hm tables, the traces' spectra, the energies.
extern "C" int emu_dc_energy_pv(int M, int nt, size_t npix, int n_bands, int n_taps, const float *in, const float *filters,
                                float *energy)
{
    if (!dc_energy_pv_supported((size_t)M, n_taps)) return -4;
    PlanHost P;
    if (!build_plan((size_t)M, P, true)) return -2;
    PlanDev D = plan_dev(P, P.tw.data(), P.tw_split.data(), nullptr, nullptr);
    if (!P.f_t1.empty()) {  // the forward transform on the F core too (k_dc_fft_f) where M has one, as deconv_api.cpp runs it
        D.f_t1 = P.f_t1.data();
        D.f_t2 = P.f_t2.data();
        D.f_w2n = P.f_w2n.data();
    }
    const int nk = M / 2 + 1, s = (n_taps - 1) / 2, gstride = dc_pv_gstride(nk);
    auto trig = [](int m_, std::vector<double> &cs, std::vector<double> &sn) {
        cs.resize((size_t)m_); sn.resize((size_t)m_);
        for (int m = 0; m < m_; ++m) {
            const double a = -2.0 * 3.14159265358979323846 * (double)m / (double)m_;
            cs[(size_t)m] = std::cos(a);
            sn[(size_t)m] = std::sin(a);
        }
    };
    std::vector<double> cs, sn, cs_e, sn_e;
    trig(M, cs, sn);
    trig(512, cs_e, sn_e);
    std::vector<c32> H((size_t)n_bands * nk), hht((size_t)2 * n_bands * 512), hpm((size_t)n_bands * 1024), spec(npix * (size_t)nk);
    std::vector<float> halves((size_t)2 * n_bands * s), g((size_t)n_bands * gstride);
    for (int b = 0; b < n_bands; ++b) {
        const float *h = filters + (size_t)b * n_taps;
        std::copy(h, h + s, halves.begin() + (size_t)(2 * b) * s);
        std::copy(h + s + 1, h + 2 * s + 1, halves.begin() + (size_t)(2 * b + 1) * s);
    }
    launch_dc_filter_spectra(nullptr, filters, n_bands, n_taps, cs.data(), sn.data(), (unsigned)M, (unsigned)nk, H.data());
    launch_dc_filter_spectra(nullptr, halves.data(), 2 * n_bands, s, cs_e.data(), sn_e.data(), 512u, 512u, hht.data());
    launch_dc_pv_tables(nullptr, n_bands, nk, gstride, (size_t)M, H.data(), hht.data(), g.data(), hpm.data());
    std::vector<c32> t1, t2;
    dc_pv_core_tables(t1, t2);
    launch_dc_fft(nullptr, D, npix, nt, in, spec.data());
    const DcPvTables T{t1.data(), t2.data(), hpm.data(), g.data(), gstride};
    launch_dc_energy_pv(nullptr, T, npix, nt, n_bands, (n_taps - 1) / 2, nk, in, spec.data(), energy);
    return 0;
}
