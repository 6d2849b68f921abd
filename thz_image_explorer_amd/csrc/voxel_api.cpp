// voxel_api.cpp — C ABI of the 3-D voxel envelope (include/thzgpu.h, K15).
#include "ctx.hpp"
#include "host_windows.hpp"

#include <cstring>

using namespace thz;

namespace {

// key <-> float of the radix select (voxel.hip sel_key)
inline float key_to_float(uint32_t key)
{
    const uint32_t bits = (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key;
    float f;
    std::memcpy(&f, &bits, sizeof f);
    return f;
}

}  // namespace

extern "C" {

int thz_voxel_cfg_default(thz_voxel_cfg *out)
{
    if (!out) return THZ_ERR_INVALID;
    out->opacity_threshold = 0.1f;  // application.rs:202-205
    out->contrast = 2.0f;
    out->sigma = 3.0f;
    out->radius = 9;
    return THZ_OK;
}

int thz_host_gaussian_kernel1d(float sigma, int radius, float *out)
{
    if (!out || radius < 0) return THZ_ERR_INVALID;
    gaussian_kernel1d(sigma, radius, out);
    return THZ_OK;
}

int thz_voxel_opacity(thz_ctx *ctx, size_t npix, size_t nt, const float *d_data, const thz_voxel_cfg *cfg,
                      float *d_opacity)
{
    if (!ctx) return THZ_ERR_INVALID;
    if (!d_data || !d_opacity || !cfg || d_data == d_opacity || cfg->radius < 0 || cfg->radius > 4096)
        return fail(ctx, THZ_ERR_INVALID, "thz_voxel_opacity: bad argument");
    if (nt < 1 || nt > (size_t)kVoxMaxNt)
        return fail(ctx, THZ_ERR_UNSUPPORTED, "thz_voxel_opacity: trace length must be 1..8192");
    if (int rc = use_device(ctx)) return rc;
    if (npix == 0) return THZ_OK;
    const int r = cfg->radius;
    std::vector<float> k((size_t)2 * r + 1);
    gaussian_kernel1d(cfg->sigma, r, k.data());
    VoxelTaps taps;
    for (int i = 0; i < kVoxTaps; ++i) taps.c[i] = 0.0f;
    const float *d_wide = nullptr;
    if (r <= kVoxPad) {
        for (int i = 0; i <= 2 * r; ++i) taps.c[i + (kVoxPad - r)] = k[(size_t)i];
    } else {
        if (int rc = ensure_ws(ctx, k.size() * sizeof(float))) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(ctx->ws, k.data(), k.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // k dies at return
        d_wide = static_cast<const float *>(ctx->ws);
    }
    StageTimer t(ctx, THZ_STAGE_VOXEL_OPACITY);
    if (!launch_voxel_opacity(ctx->stream, npix, (int)nt, d_data, taps, d_wide, r, cfg->contrast,
                              cfg->opacity_threshold, d_opacity))
        return fail(ctx, THZ_ERR_UNSUPPORTED, "thz_voxel_opacity: no kernel for this shape");
    return check_launch(ctx);
}

int thz_select_histogram(thz_ctx *ctx, const float *d_vals, size_t n, int level, uint32_t prefix, uint64_t *d_hist)
{
    if (!ctx) return THZ_ERR_INVALID;
    if (!d_vals || !d_hist || level < 0 || level > 2 || (uintptr_t)d_vals % 16 != 0)
        return fail(ctx, THZ_ERR_INVALID, "thz_select_histogram: bad argument (values must be 16-byte aligned)");
    if (int rc = use_device(ctx)) return rc;
    if (n == 0) return THZ_OK;
    launch_select_hist(ctx->stream, d_vals, n, level, prefix, reinterpret_cast<unsigned long long *>(d_hist));
    return check_launch(ctx);
}

int thz_host_select_step(const uint64_t *hist, int nbins, uint64_t k, int *bin, uint64_t *k_rem)
{
    if (!hist || !bin || !k_rem || nbins < 1 || k < 1) return THZ_ERR_INVALID;
    unsigned long long rem = 0;
    if (select_step(reinterpret_cast<const unsigned long long *>(hist), nbins, k, bin, &rem)) return THZ_ERR_INVALID;
    *k_rem = rem;
    return THZ_OK;
}

float thz_host_select_value(int bin0, int bin1, int bin2)
{
    return key_to_float(((uint32_t)bin0 << 21) | ((uint32_t)bin1 << 10) | (uint32_t)bin2);
}

int thz_kth_largest(thz_ctx *ctx, const float *d_vals, size_t n, uint64_t k, float *out)
{
    if (!ctx) return THZ_ERR_INVALID;
    if (!d_vals || !out || k < 1 || k > n) return fail(ctx, THZ_ERR_INVALID, "thz_kth_largest: need 1 <= k <= n");
    if (int rc = use_device(ctx)) return rc;
    if (int rc = ensure_ws(ctx, kSelBins * sizeof(uint64_t))) return rc;
    uint64_t *d_hist = static_cast<uint64_t *>(ctx->ws);
    std::vector<uint64_t> hist(kSelBins);
    StageTimer t(ctx, THZ_STAGE_VOXEL_SELECT);
    int bins[3] = {0, 0, 0};
    uint32_t prefix = 0;
    uint64_t rank = k;
    // level 0 first with everything below 2^-10 lumped into one bin (cheap: long runs); the
    // full histogram only if the k-th largest turns out to be that small
    uint32_t floor_bin = (0x80000000u | 0x3A800000u) >> 21;
    for (int level = 0; level < 3; ++level) {
        for (;;) {
            HIP_TRY(ctx, hipMemsetAsync(d_hist, 0, kSelBins * sizeof(uint64_t), ctx->stream));
            if (int rc = thz_select_histogram(ctx, d_vals, n, level, level == 0 ? floor_bin : prefix, d_hist)) return rc;
            HIP_TRY(ctx, hipMemcpyAsync(hist.data(), d_hist, kSelBins * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            uint64_t r = 0;
            if (thz_host_select_step(hist.data(), level == 2 ? 1024 : kSelBins, rank, &bins[level], &r))
                return fail(ctx, THZ_ERR_INVALID, "thz_kth_largest: histogram holds fewer than k values");
            if (level == 0 && floor_bin != 0 && (uint32_t)bins[0] == floor_bin) {
                floor_bin = 0;
                continue;
            }
            rank = r;
            break;
        }
        prefix = level == 0 ? (uint32_t)bins[0] : (((uint32_t)bins[0] << 11) | (uint32_t)bins[1]);
    }
    *out = thz_host_select_value(bins[0], bins[1], bins[2]);
    return THZ_OK;
}

int thz_voxel_threshold(thz_ctx *ctx, const float *d_opacity, size_t n, uint64_t max_instances, float *out)
{
    if (!ctx || !out) return THZ_ERR_INVALID;
    if (max_instances < 1) return fail(ctx, THZ_ERR_INVALID, "thz_voxel_threshold: max_instances must be >= 1");
    if (n <= max_instances) {
        *out = 0.0f;
        return THZ_OK;
    }
    return thz_kth_largest(ctx, d_opacity, n, max_instances, out);
}

int thz_voxel_instances(thz_ctx *ctx, const float *d_opacity, size_t gw, size_t gh, size_t gd, size_t x0,
                        size_t gw_total, float threshold, float time_span, int scaling, size_t orig_w,
                        size_t orig_h, size_t orig_d, thz_voxel_instance *d_out, uint64_t capacity,
                        uint64_t *count, float *cube_dims)
{
    if (!ctx) return THZ_ERR_INVALID;
    if (!d_opacity || !count || (!d_out && capacity) || gh == 0 || gd == 0 || gd > ((size_t)1 << 24)
        || x0 + gw > gw_total || (uintptr_t)d_out % 16 != 0)
        return fail(ctx, THZ_ERR_INVALID, "thz_voxel_instances: bad argument");
    if (int rc = use_device(ctx)) return rc;
    const VoxelLayout L = voxel_layout(time_span, gw_total, gh, gd, orig_w, orig_h, orig_d);
    if (cube_dims) {
        cube_dims[0] = L.cube_width;
        cube_dims[1] = L.cube_height;
        cube_dims[2] = L.cube_depth;
    }
    *count = 0;
    const size_t npix = gw * gh;
    if (npix == 0) return THZ_OK;
    // workspace: counts u32[npix] | offsets u64[npix] | tile sums u64[ntiles] | total u64
    const size_t ntiles = (npix + 2047) / 2048;
    const size_t off_counts = 0;
    const size_t off_offsets = (npix * sizeof(uint32_t) + 15) & ~(size_t)15;
    const size_t off_tiles = off_offsets + npix * sizeof(uint64_t);
    const size_t off_total = off_tiles + ntiles * sizeof(uint64_t);
    if (int rc = ensure_ws(ctx, off_total + sizeof(uint64_t))) return rc;
    char *ws = static_cast<char *>(ctx->ws);
    uint32_t *d_counts = reinterpret_cast<uint32_t *>(ws + off_counts);
    unsigned long long *d_offsets = reinterpret_cast<unsigned long long *>(ws + off_offsets);
    unsigned long long *d_tiles = reinterpret_cast<unsigned long long *>(ws + off_tiles);
    unsigned long long *d_total = reinterpret_cast<unsigned long long *>(ws + off_total);
    VoxelGeom g{L.spacing_w, L.spacing_h, L.spacing_d, L.half_w, L.half_h, L.half_d, (float)scaling, threshold, x0};
    StageTimer t(ctx, THZ_STAGE_VOXEL_EMIT);
    launch_voxel_count(ctx->stream, npix, (int)gd, d_opacity, threshold, d_counts);
    launch_scan_counts(ctx->stream, d_counts, npix, d_tiles, d_offsets, d_total);
    launch_voxel_emit(ctx->stream, npix, (int)gd, gh, d_opacity, d_counts, d_offsets, g, reinterpret_cast<float *>(d_out),
                      capacity);
    if (int rc = check_launch(ctx)) return rc;
    unsigned long long total = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&total, d_total, sizeof total, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *count = total;
    return THZ_OK;
}

int thz_traffic_probe(thz_ctx *ctx, size_t npix, size_t nt, const float *d_in, float *d_fft, float *d_amp,
                      float *d_phase, float *d_data_out)
{
    if (!ctx) return THZ_ERR_INVALID;
    if (!d_in || !d_fft || !d_amp || !d_phase || !d_data_out || nt < 8 || nt % 8 != 0 || nt > (1u << 20))
        return fail(ctx, THZ_ERR_INVALID, "thz_traffic_probe: bad argument");
    if (int rc = use_device(ctx)) return rc;
    if (npix == 0) return THZ_OK;
    StageTimer t(ctx, THZ_STAGE_PROBE);
    launch_traffic_probe(ctx->stream, npix, (int)nt, d_in, d_fft, d_amp, d_phase, d_data_out);
    return check_launch(ctx);
}

}  // extern "C"
