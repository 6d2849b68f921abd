// voxel.hip — K15: the 3-D voxel envelope the data thread rebuilds after every
// recompute (update_intensity_image, data_thread.rs:48-101 ->
// instance_from_data, gui/threed_plot.rs:132-276).
//
//   k_voxel_opacity   per trace: (v^2)^contrast, 1-D Gaussian, max/min rule     (HBM: 8 nt B/trace)
//   k_select_hist     radix select of the max_instances-th largest opacity:
//                     three 11/11/10-bit histogram levels over the cube          (HBM: 4 B/voxel/level)
//   k_voxel_count / k_scan_* / k_voxel_emit
//                     ordered compaction of the voxels >= threshold into
//                     InstanceData records (position, scale, colour)             (HBM: 4 B/voxel each)
//
// Built with -ffp-contract=off: instance positions follow the reference's separate
// multiply and subtract (HIP's __fmul_rn is a plain product and would still fuse);
// the convolution asks for its FMAs explicitly.
//
// All of it is HBM-bound streaming over the (nx, ny, nt) f32 cube; one wave owns
// one trace at a time like every other kernel of the engine.
#include "kernels.hpp"
#include "thz_device.hpp"

#include <math.h>
#include <stdlib.h>

namespace thz {

#ifndef THZ_EMU
__device__ __forceinline__ uint64_t wave_ballot(bool p) { return __ballot(p); }
__device__ __forceinline__ uint32_t wave_read_lane_u32(uint32_t v, int lane_uniform)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, lane_uniform);
}
__device__ __forceinline__ float hw_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float hw_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float sub_rn(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float div_rn(float a, float b) { return __fdiv_rn(a, b); }
__device__ __forceinline__ int popc64(uint64_t m) { return __popcll(m); }
__device__ __forceinline__ int ffs64(uint64_t m) { return __ffsll((unsigned long long)m); }
#else
inline uint64_t wave_ballot(bool p)
{
    uint64_t m = 0;
    for (int i = 0; i < 64; ++i)
        if (wave_shfl(p ? 1.0f : 0.0f, i) != 0.0f) m |= (uint64_t)1 << i;
    return m;
}
inline uint32_t wave_read_lane_u32(uint32_t v, int lane)
{
    return __builtin_bit_cast(uint32_t, wave_shfl(__builtin_bit_cast(float, v), lane));
}
inline float hw_exp2(float x) { return exp2f(x); }
inline float hw_log2(float x) { return log2f(x); }
inline float mul_rn(float a, float b) { return a * b; }
inline float sub_rn(float a, float b) { return a - b; }
inline float div_rn(float a, float b) { return a / b; }
inline int popc64(uint64_t m) { return __builtin_popcountll(m); }
inline int ffs64(uint64_t m) { return __builtin_ffsll((long long)m); }
static inline unsigned int atomicAdd(unsigned int *p, unsigned int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v)
{
    return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}
#endif

typedef float f2 __attribute__((ext_vector_type(2)));

// (a.hi, b.lo) as one register pair: v_pk_mov_b32 picks a half of each source
__device__ __forceinline__ f2 pair_hi_lo(f2 a, f2 b)
{
#ifndef THZ_EMU
    f2 r;
    asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return f2{a.y, b.x};
#endif
}

__device__ __forceinline__ float wave_max(float v)
{
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, wave_shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ float wave_min(float v)
{
    for (int m = 32; m >= 1; m >>= 1) v = fminf(v, wave_shfl_xor(v, m));
    return v;
}

// ------------------------------------------------------------------ opacity
// (v^2)^contrast: powi(2) then powf(contrast), threed_plot.rs:169 and :113.
// pow_mode 1 / 2: contrast is exactly 1 / 2 (the default) -> plain products.
__device__ __forceinline__ float vox_pow(float v, float contrast, int pow_mode)
{
    const float q = v * v;
    if (pow_mode == 2) return q * q;
    if (pow_mode == 1) return q;
    return hw_exp2(contrast * hw_log2(q));  // q = 0 -> exp2(-inf) = 0
}

// n / d with inv = 1/d: quotient estimate plus one residual correction — the
// result of the IEEE division for all but rare last-place cases, and exactly 1
// for n == d, at 3 instructions instead of the ~10 of the full sequence.
__device__ __forceinline__ float vox_div(float n, float d, float inv)
{
    const float q = n * inv;
    return fmaf(fmaf(-d, q, n), inv, q);
}

template <int NQ, bool VEC>
__device__ __forceinline__ void vox_load(const float *__restrict__ src, int nt, int lane, float4 (&cur)[NQ])
{
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = q * 256 + 4 * lane;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (VEC) {
            if (e < nt) v = *reinterpret_cast<const float4 *>(src + e);
        } else {
            if (e + 0 < nt) v.x = src[e + 0];
            if (e + 1 < nt) v.y = src[e + 1];
            if (e + 2 < nt) v.z = src[e + 2];
            if (e + 3 < nt) v.w = src[e + 3];
        }
        cur[q] = v;
    }
}

// One wave per trace.  The powered samples go to the wave's LDS slice
// [pad zeros | NQ*256 samples (zeros past nt) | pad zeros]; every lane then owns
// the quads e = 256 q + 4 lane.
//  * WIDE = false (radius <= kVoxPad): the 4 + 2*kVoxPad samples around a quad are
//    read once with seven ds_read_b128 and the kernel runs over a zero-padded
//    25-tap vector held in SGPRs — adding tap*0 keeps the reference's k-ascending
//    summation of the real taps;
//  * WIDE = true: any radius, one LDS read per tap and output.
// Out-of-range taps are skipped by the reference (threed_plot.rs:112); reading a
// zero adds +0.
template <int NQ, bool VEC, bool WIDE>
__global__ __launch_bounds__(256) void k_voxel_opacity(size_t npix, int nt, const float *__restrict__ data,
                                                       VoxelTaps taps, const float *__restrict__ wide_taps,
                                                       int radius, int pad, float contrast, int pow_mode,
                                                       float opacity_threshold, float *__restrict__ out)
{
    constexpr bool PREFETCH = NQ <= 16;
    THZ_DYN_LDS(smem);
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6), wpb = (int)(blockDim.x >> 6);
    const int padc = WIDE ? (pad & ~3) : kVoxPad;  // multiple of 4: every quad access is 16-byte aligned
    const int slice = NQ * 256 + 2 * padc;
    float *lds = static_cast<float *>(__builtin_assume_aligned(
        reinterpret_cast<float *>(smem) + (size_t)wave * slice, 16));
    for (int i = lane; i < slice; i += kWave) lds[i] = 0.0f;
    wave_sync();

    const size_t stride = (size_t)gridDim.x * wpb;
    size_t trace = (size_t)blockIdx.x * wpb + wave;
    float4 cur[NQ];
    if (trace < npix) vox_load<NQ, VEC>(data + trace * (size_t)nt, nt, lane, cur);
    for (; trace < npix; trace += stride) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = q * 256 + 4 * lane;
            float4 p;
            p.x = vox_pow(cur[q].x, contrast, pow_mode);
            p.y = vox_pow(cur[q].y, contrast, pow_mode);
            p.z = vox_pow(cur[q].z, contrast, pow_mode);
            p.w = vox_pow(cur[q].w, contrast, pow_mode);
            *static_cast<float4 *>(__builtin_assume_aligned(lds + padc + e, 16)) = p;  // samples past nt were loaded as 0
        }
        wave_sync();
        const size_t next = trace + stride;
        if (PREFETCH && next < npix) vox_load<NQ, VEC>(data + next * (size_t)nt, nt, lane, cur);

        float4 o[NQ];
        float mx = -INFINITY, mn = INFINITY;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = q * 256 + 4 * lane;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            if constexpr (!WIDE) {
                // lds[e + j] = p[e + j - kVoxPad], j = 0 .. 27, as 14 register pairs P[m] =
                // (w[2m], w[2m+1]) plus the 13 odd-aligned pairs S[m] = (w[2m+1], w[2m+2]), so
                // that every tap is two packed FMAs: (a0,a1) += (w[k],w[k+1]) c, (a2,a3) += (w[k+2],w[k+3]) c
                constexpr int NP = (4 + 2 * kVoxPad) / 2;
                f2 P[NP], S[NP - 1];
#pragma unroll
                for (int j = 0; j < NP / 2; ++j) {
                    const float4 t = *static_cast<const float4 *>(__builtin_assume_aligned(lds + e + 4 * j, 16));
                    P[2 * j] = f2{t.x, t.y};
                    P[2 * j + 1] = f2{t.z, t.w};
                }
#pragma unroll
                for (int m = 0; m < NP - 1; ++m) S[m] = pair_hi_lo(P[m], P[m + 1]);
                f2 a01 = f2{0.f, 0.f}, a23 = f2{0.f, 0.f};
#pragma unroll
                for (int k = 0; k < kVoxTaps; ++k) {
                    const f2 c = f2{taps.c[k], taps.c[k]};
                    if (k % 2 == 0) {
                        a01 = __builtin_elementwise_fma(P[k / 2], c, a01);
                        a23 = __builtin_elementwise_fma(P[k / 2 + 1], c, a23);
                    } else {
                        a01 = __builtin_elementwise_fma(S[k / 2], c, a01);
                        a23 = __builtin_elementwise_fma(S[k / 2 + 1], c, a23);
                    }
                }
                a0 = a01.x; a1 = a01.y; a2 = a23.x; a3 = a23.y;
            } else {
                const float *w = lds + padc + e - radius;  // w[k + o] = p[e + o + k - radius]
                for (int k = 0; k <= 2 * radius; ++k) {
                    const float c = wide_taps[k];
                    a0 = fmaf(w[k + 0], c, a0);
                    a1 = fmaf(w[k + 1], c, a1);
                    a2 = fmaf(w[k + 2], c, a2);
                    a3 = fmaf(w[k + 3], c, a3);
                }
            }
            if (e + 0 < nt) { mx = fmaxf(mx, a0); mn = fminf(mn, a0); }
            if (e + 1 < nt) { mx = fmaxf(mx, a1); mn = fminf(mn, a1); }
            if (e + 2 < nt) { mx = fmaxf(mx, a2); mn = fminf(mn, a2); }
            if (e + 3 < nt) { mx = fmaxf(mx, a3); mn = fminf(mn, a3); }
            o[q] = make_float4(a0, a1, a2, a3);
        }
        mx = wave_max(mx);
        mn = wave_min(mn);
        // threed_plot.rs:181-199: whole line zero below the opacity threshold or when flat
        const bool keep = !(mx < opacity_threshold) && fabsf(mx - mn) > 1e-6f;
        const float range = mx - mn;
        const float inv = div_rn(1.0f, range);
        float *dst = out + trace * (size_t)nt;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = q * 256 + 4 * lane;
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (keep) {
                r.x = vox_div(o[q].x - mn, range, inv);
                r.y = vox_div(o[q].y - mn, range, inv);
                r.z = vox_div(o[q].z - mn, range, inv);
                r.w = vox_div(o[q].w - mn, range, inv);
            }
            if constexpr (VEC) {
                if (e < nt) *reinterpret_cast<float4 *>(dst + e) = r;
            } else {
                if (e + 0 < nt) dst[e + 0] = r.x;
                if (e + 1 < nt) dst[e + 1] = r.y;
                if (e + 2 < nt) dst[e + 2] = r.z;
                if (e + 3 < nt) dst[e + 3] = r.w;
            }
        }
        wave_sync();  // the slice is rewritten by the next trace
        if (!PREFETCH && next < npix) vox_load<NQ, VEC>(data + next * (size_t)nt, nt, lane, cur);
    }
}

template <int NQ, bool VEC, bool WIDE>
static void launch_voxel_opacity_t(hipStream_t st, size_t npix, int nt, const float *data, const VoxelTaps &taps,
                                   const float *wide_taps, int radius, float contrast, int pow_mode,
                                   float opacity_threshold, float *out)
{
    const int pad = WIDE ? ((radius + 3) & ~3) : kVoxPad;
    const int wpb = 4;
    const size_t lds = (size_t)wpb * (NQ * 256 + 2 * pad) * sizeof(float);
    size_t blocks = (npix + wpb - 1) / wpb;
    // persistent-ish grid: enough blocks for every CU's LDS, the waves stride over the traces
    const size_t per_cu = lds ? (160 * 1024) / lds : 8;
    const size_t cap = 256 * (per_cu ? (per_cu > 8 ? 8 : per_cu) : 1);
    if (blocks > cap) blocks = cap;
    if (blocks == 0) return;
    THZ_LAUNCH((k_voxel_opacity<NQ, VEC, WIDE>), blocks, wpb * kWave, lds, st, npix, nt, data, taps, wide_taps,
               radius, pad, contrast, pow_mode, opacity_threshold, out);
}

template <bool VEC, bool WIDE>
static bool launch_voxel_opacity_q(hipStream_t st, size_t npix, int nt, const float *data, const VoxelTaps &taps,
                                   const float *wide_taps, int radius, float contrast, int pow_mode,
                                   float opacity_threshold, float *out)
{
    const int nq = (nt + 255) / 256;
#define THZ_VOX_CASE(N)                                                                                      \
    if (nq <= N) {                                                                                           \
        launch_voxel_opacity_t<N, VEC, WIDE>(st, npix, nt, data, taps, wide_taps, radius, contrast, pow_mode, \
                                             opacity_threshold, out);                                        \
        return true;                                                                                         \
    }
    THZ_VOX_CASE(1)
    THZ_VOX_CASE(2)
    THZ_VOX_CASE(4)
    THZ_VOX_CASE(8)
    THZ_VOX_CASE(16)
    THZ_VOX_CASE(32)
#undef THZ_VOX_CASE
    return false;
}

bool launch_voxel_opacity(hipStream_t st, size_t npix, int nt, const float *data, const VoxelTaps &taps,
                          const float *wide_taps, int radius, float contrast, float opacity_threshold,
                          float *out)
{
    if (nt < 1 || nt > kVoxMaxNt) return false;
    const int pow_mode = contrast == 2.0f ? 2 : (contrast == 1.0f ? 1 : 0);
    const bool vec = (nt % 4 == 0) && ((uintptr_t)data % 16 == 0) && ((uintptr_t)out % 16 == 0);
    const bool wide = radius > kVoxPad;
    if (wide && !wide_taps) return false;
    if (vec)
        return wide ? launch_voxel_opacity_q<true, true>(st, npix, nt, data, taps, wide_taps, radius, contrast,
                                                         pow_mode, opacity_threshold, out)
                    : launch_voxel_opacity_q<true, false>(st, npix, nt, data, taps, wide_taps, radius, contrast,
                                                          pow_mode, opacity_threshold, out);
    return wide ? launch_voxel_opacity_q<false, true>(st, npix, nt, data, taps, wide_taps, radius, contrast,
                                                      pow_mode, opacity_threshold, out)
                : launch_voxel_opacity_q<false, false>(st, npix, nt, data, taps, wide_taps, radius, contrast,
                                                       pow_mode, opacity_threshold, out);
}

// ------------------------------------------------------------- radix select
// Order-preserving key of an f32: larger float <=> larger key (negative values
// included, although opacities are never negative).
__device__ __forceinline__ uint32_t sel_key(uint32_t bits)
{
    return (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
}

// Keys into the block's LDS histogram.  Neighbouring samples of a smooth
// envelope (and the many exact zeros) fall into the same bin, which would
// serialise a plain LDS atomic 64 ways.
// level 0 lumps every key below bin `prefix` (the "floor") into that bin: the k-th
// largest of a cube sits near the top, and the lump turns the tails and zero lines
// into long runs.  The caller repeats level 0 with floor 0 if the walk ends in the lump.
__device__ __forceinline__ uint32_t sel_bin(uint32_t key, int level, uint32_t prefix)
{
    if (level == 0) {
        const uint32_t b = key >> 21;
        return b > prefix ? b : prefix;
    }
    return level == 1 ? (key >> 10) & 2047u : key & 1023u;
}
__device__ __forceinline__ bool sel_match(uint32_t key, int level, uint32_t prefix)
{
    return level == 0 ? true : (level == 1 ? (key >> 21) == prefix : (key >> 10) == prefix);
}

// The 64 lanes hold consecutive (stride-4) samples, so equal bins come in runs:
// the first lane of each run adds the run's length.  One step, no loop; a noisy
// wave degenerates to one atomic per lane.
__device__ __forceinline__ void sel_add(unsigned int *h, uint32_t bin, bool active, unsigned int weight)
{
    const int lane = lane_id();
    const uint32_t tag = active ? bin : 0xFFFFFFFFu;  // no bin has this value
    const uint32_t prev = __builtin_bit_cast(uint32_t, wave_shr1(__builtin_bit_cast(float, tag)));
    const bool head = lane == 0 || tag != prev;
    const uint64_t heads = wave_ballot(head);
    if (active && head) {
        const uint64_t above = (heads >> lane) >> 1;
        const int len = above ? __builtin_ctzll(above) + 1 : kWave - lane;
        atomicAdd(&h[bin], (unsigned int)len * weight);
    }
}

// four consecutive values of one lane: when they share a bin in every lane of the wave
// (zero lines, smooth envelopes) one weighted round does the work of four
__device__ __forceinline__ void sel_add4(unsigned int *h, uint4 k, bool in, int level, uint32_t prefix)
{
    const uint32_t kx = sel_key(k.x), ky = sel_key(k.y), kz = sel_key(k.z), kw = sel_key(k.w);
    const uint32_t bx = sel_bin(kx, level, prefix), by = sel_bin(ky, level, prefix);
    const uint32_t bz = sel_bin(kz, level, prefix), bw = sel_bin(kw, level, prefix);
    const bool ax = in && sel_match(kx, level, prefix), ay = in && sel_match(ky, level, prefix);
    const bool az = in && sel_match(kz, level, prefix), aw = in && sel_match(kw, level, prefix);
    const bool any = ax || ay || az || aw;
    if (wave_ballot(any) == 0) return;
    const bool uniform4 = !any || (ax && ay && az && aw && bx == by && by == bz && bz == bw);
    if (wave_ballot(!uniform4) == 0) {
        sel_add(h, bx, any, 4u);
    } else {
        sel_add(h, bx, ax, 1u);
        sel_add(h, by, ay, 1u);
        sel_add(h, bz, az, 1u);
        sel_add(h, bw, aw, 1u);
    }
}

// hist[bin] += number of keys of this level's bin among the values whose higher
// bits equal `prefix` (level 0: all values, `prefix` is the floor bin).  hist has
// kSelBins entries.
__global__ __launch_bounds__(256) void k_select_hist(const uint32_t *__restrict__ vals, size_t n, int level,
                                                     uint32_t prefix, unsigned long long *__restrict__ hist)
{
    __shared__ unsigned int h[kSelBins];
    for (int i = (int)threadIdx.x; i < kSelBins; i += (int)blockDim.x) h[i] = 0u;
    __syncthreads();
    constexpr int U = 4;           // loads in flight per lane
    constexpr size_t kChunk = 1024;  // quads per wave step: 16 KB contiguous, like a trace
    const size_t nq = n / 4;
    const int lane = lane_id();
    const size_t wave_id = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const size_t n_waves = (size_t)gridDim.x * (blockDim.x >> 6);
    const uint4 *v4 = reinterpret_cast<const uint4 *>(vals);
    for (size_t c = wave_id; c * kChunk < nq; c += n_waves) {  // wave-uniform trip counts: ballots stay uniform
        for (int it = 0; it < (int)(kChunk / (U * kWave)); ++it) {
            uint4 k[U];
            bool in[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t q = c * kChunk + (size_t)(it * U + u) * kWave + lane;
                in[u] = q < nq;
                k[u] = make_uint4(0u, 0u, 0u, 0u);
                if (in[u]) k[u] = v4[q];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) sel_add4(h, k[u], in[u], level, prefix);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < kWave) {  // the n % 4 tail
        const size_t i = nq * 4 + threadIdx.x;
        const bool in = i < n;
        const uint32_t key = sel_key(in ? vals[i] : 0u);
        sel_add(h, sel_bin(key, level, prefix), in && sel_match(key, level, prefix), 1u);
    }
    __syncthreads();
    for (int i = (int)threadIdx.x; i < kSelBins; i += (int)blockDim.x)
        if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

void launch_select_hist(hipStream_t st, const float *vals, size_t n, int level, uint32_t prefix,
                        unsigned long long *hist)
{
    size_t blocks = (n / 4 + 4095) / 4096;  // 4 waves x 1024 quads
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    THZ_LAUNCH(k_select_hist, blocks, 256, 0, st, reinterpret_cast<const uint32_t *>(vals), n, level, prefix, hist);
}

// ------------------------------------------------- ordered compaction: counts
template <bool VEC>
__global__ __launch_bounds__(256) void k_voxel_count(size_t npix, int nt, const float *__restrict__ opacity,
                                                     float threshold, uint32_t *__restrict__ counts)
{
    const int lane = lane_id();
    const int wpb = (int)(blockDim.x >> 6);
    const size_t stride = (size_t)gridDim.x * wpb;
    for (size_t trace = (size_t)blockIdx.x * wpb + (threadIdx.x >> 6); trace < npix; trace += stride) {
        const float *src = opacity + trace * (size_t)nt;
        float c = 0.0f;  // <= nt < 2^24: exact
        if constexpr (VEC) {
            for (int e = 4 * lane; e < nt; e += 256) {
                const float4 v = *reinterpret_cast<const float4 *>(src + e);
                c += (v.x >= threshold ? 1.0f : 0.0f) + (v.y >= threshold ? 1.0f : 0.0f)
                     + (v.z >= threshold ? 1.0f : 0.0f) + (v.w >= threshold ? 1.0f : 0.0f);
            }
        } else {
            for (int e = lane; e < nt; e += kWave) c += src[e] >= threshold ? 1.0f : 0.0f;
        }
        c = wave_reduce_add(c);
        if (lane == 0) counts[trace] = (uint32_t)c;
    }
}

void launch_voxel_count(hipStream_t st, size_t npix, int nt, const float *opacity, float threshold,
                        uint32_t *counts)
{
    size_t blocks = (npix + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks == 0) return;
    if (nt % 4 == 0 && (uintptr_t)opacity % 16 == 0)
        THZ_LAUNCH(k_voxel_count<true>, blocks, 256, 0, st, npix, nt, opacity, threshold, counts);
    else
        THZ_LAUNCH(k_voxel_count<false>, blocks, 256, 0, st, npix, nt, opacity, threshold, counts);
}

// ---------------------------------------- exclusive scan counts(u32) -> offsets(u64)
constexpr int kScanTile = 2048;  // counts per block: 256 threads x 8

// exclusive scan of one value per thread over the block; returns the block total
__device__ __forceinline__ unsigned long long block_excl_scan(unsigned long long v, unsigned long long *sh,
                                                              unsigned long long *total)
{
    const int t = (int)threadIdx.x, n = (int)blockDim.x;
    sh[t] = v;
    __syncthreads();
    for (int d = 1; d < n; d <<= 1) {
        const unsigned long long add = t >= d ? sh[t - d] : 0ull;
        __syncthreads();
        sh[t] += add;
        __syncthreads();
    }
    const unsigned long long incl = sh[t];
    *total = sh[n - 1];
    __syncthreads();
    return incl - v;
}

__global__ __launch_bounds__(256) void k_scan_tile_sums(const uint32_t *__restrict__ counts, size_t n,
                                                        unsigned long long *__restrict__ tile_sums)
{
    __shared__ unsigned long long sh[256];
    const size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * 8;
    unsigned long long s = 0;
    for (int j = 0; j < 8; ++j)
        if (base + j < n) s += counts[base + j];
    unsigned long long total;
    (void)block_excl_scan(s, sh, &total);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// single block: tile_sums -> exclusive offsets in place, grand total to *total
__global__ __launch_bounds__(256) void k_scan_tile_offsets(unsigned long long *__restrict__ tile_sums, size_t ntiles,
                                                           unsigned long long *__restrict__ total)
{
    __shared__ unsigned long long sh[256];
    unsigned long long carry = 0;
    for (size_t base = 0; base < ntiles; base += blockDim.x) {
        const size_t i = base + threadIdx.x;
        const unsigned long long v = i < ntiles ? tile_sums[i] : 0ull;
        unsigned long long chunk;
        const unsigned long long ex = block_excl_scan(v, sh, &chunk);
        if (i < ntiles) tile_sums[i] = carry + ex;
        carry += chunk;
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(256) void k_scan_apply(const uint32_t *__restrict__ counts, size_t n,
                                                    const unsigned long long *__restrict__ tile_offsets,
                                                    unsigned long long *__restrict__ offsets)
{
    __shared__ unsigned long long sh[256];
    const size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * 8;
    uint32_t c[8];
    unsigned long long s = 0;
    for (int j = 0; j < 8; ++j) {
        c[j] = base + j < n ? counts[base + j] : 0u;
        s += c[j];
    }
    unsigned long long total;
    unsigned long long run = tile_offsets[blockIdx.x] + block_excl_scan(s, sh, &total);
    for (int j = 0; j < 8; ++j) {
        if (base + j < n) offsets[base + j] = run;
        run += c[j];
    }
}

// counts[n] -> offsets[n] (exclusive), *total; tile_ws holds ceil(n / kScanTile) u64
void launch_scan_counts(hipStream_t st, const uint32_t *counts, size_t n, unsigned long long *tile_ws,
                        unsigned long long *offsets, unsigned long long *total)
{
    const size_t ntiles = (n + kScanTile - 1) / kScanTile;
    if (ntiles == 0) return;
    THZ_LAUNCH(k_scan_tile_sums, ntiles, 256, 0, st, counts, n, tile_ws);
    THZ_LAUNCH(k_scan_tile_offsets, 1, 256, 0, st, tile_ws, ntiles, total);
    THZ_LAUNCH(k_scan_apply, ntiles, 256, 0, st, counts, n, tile_ws, offsets);
}

// ------------------------------------------------------------------- emit
__device__ __forceinline__ float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

// bevy_color Srgba -> LinearRgba channel conversion (oracle/thz_oracle_voxel.c header)
__device__ __forceinline__ float srgb_to_linear(float x)
{
    if (x <= 0.0f) return x;
    if (x <= 0.04045f) return div_rn(x, 12.92f);
    return powf(div_rn(x + 0.055f, 1.055f), 2.4f);
}

__device__ __forceinline__ void vox_emit_one(float4 *__restrict__ out, unsigned long long pos, float px, float py,
                                             int z, float o, const VoxelGeom &g)
{
    const float v = div_rn(o - g.threshold, 1.0f - g.threshold);
    const float four = mul_rn(4.0f, v);
    const float r = clamp01(four - 1.5f);
    const float gg = clamp01(four - 0.5f) - clamp01(four - 2.5f);
    const float b = 1.0f - clamp01(four - 1.5f);
    const float pz = sub_rn(g.half_d, mul_rn((float)z, g.spacing_d));
    out[2 * pos + 0] = make_float4(px, py, pz, g.scale);
    out[2 * pos + 1] = make_float4(srgb_to_linear(r), srgb_to_linear(gg), srgb_to_linear(b), o);
}

// Instance loop of threed_plot.rs:221-271 in its x, y, z order: trace by trace
// (traces without a voxel are not read again), inside a trace 256 samples at a
// time — a lane owns four consecutive samples, four ballots give its offset.
template <bool VEC>
__global__ __launch_bounds__(256) void k_voxel_emit(size_t npix, int nt, size_t gh, const float *__restrict__ opacity,
                                                    const uint32_t *__restrict__ counts,
                                                    const unsigned long long *__restrict__ offsets, VoxelGeom g,
                                                    float4 *__restrict__ out, unsigned long long capacity)
{
    const int lane = lane_id();
    const int wpb = (int)(blockDim.x >> 6);
    const size_t stride = (size_t)gridDim.x * wpb;
    const uint64_t below = ((uint64_t)1 << lane) - 1;
    for (size_t trace = (size_t)blockIdx.x * wpb + (threadIdx.x >> 6); trace < npix; trace += stride) {
        if (counts[trace] == 0u) continue;
        const size_t x = g.x0 + trace / gh, y = trace % gh;
        const float px = sub_rn(mul_rn((float)y, g.spacing_h), g.half_h);
        const float py = sub_rn(g.half_w, mul_rn((float)x, g.spacing_w));
        const float *src = opacity + trace * (size_t)nt;
        unsigned long long off = offsets[trace];
        for (int z0 = 0; z0 < nt; z0 += 4 * kWave) {
            const int z = z0 + 4 * lane;
            float4 o = make_float4(-1.f, -1.f, -1.f, -1.f);
            if constexpr (VEC) {
                if (z < nt) o = *reinterpret_cast<const float4 *>(src + z);
            } else {
                if (z + 0 < nt) o.x = src[z + 0];
                if (z + 1 < nt) o.y = src[z + 1];
                if (z + 2 < nt) o.z = src[z + 2];
                if (z + 3 < nt) o.w = src[z + 3];
            }
            const bool p0 = z + 0 < nt && o.x >= g.threshold, p1 = z + 1 < nt && o.y >= g.threshold;
            const bool p2 = z + 2 < nt && o.z >= g.threshold, p3 = z + 3 < nt && o.w >= g.threshold;
            const uint64_t m0 = wave_ballot(p0), m1 = wave_ballot(p1), m2 = wave_ballot(p2), m3 = wave_ballot(p3);
            if ((m0 | m1 | m2 | m3) == 0) continue;
            unsigned long long pos = off + (unsigned long long)(popc64(m0 & below) + popc64(m1 & below)
                                                                + popc64(m2 & below) + popc64(m3 & below));
            if (p0) { if (pos < capacity) vox_emit_one(out, pos, px, py, z + 0, o.x, g); ++pos; }
            if (p1) { if (pos < capacity) vox_emit_one(out, pos, px, py, z + 1, o.y, g); ++pos; }
            if (p2) { if (pos < capacity) vox_emit_one(out, pos, px, py, z + 2, o.z, g); ++pos; }
            if (p3) { if (pos < capacity) vox_emit_one(out, pos, px, py, z + 3, o.w, g); ++pos; }
            off += (unsigned long long)(popc64(m0) + popc64(m1) + popc64(m2) + popc64(m3));
        }
    }
}

void launch_voxel_emit(hipStream_t st, size_t npix, int nt, size_t gh, const float *opacity, const uint32_t *counts,
                       const unsigned long long *offsets, const VoxelGeom &g, float *out,
                       unsigned long long capacity)
{
    size_t blocks = (npix + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks == 0) return;
    if (nt % 4 == 0 && (uintptr_t)opacity % 16 == 0)
        THZ_LAUNCH(k_voxel_emit<true>, blocks, 256, 0, st, npix, nt, gh, opacity, counts, offsets, g,
                   reinterpret_cast<float4 *>(out), capacity);
    else
        THZ_LAUNCH(k_voxel_emit<false>, blocks, 256, 0, st, npix, nt, gh, opacity, counts, offsets, g,
                   reinterpret_cast<float4 *>(out), capacity);
}

// ---------------------------------------------------------------- traffic probe
// Measurement aid, not a stage: moves exactly the bytes of the fused chain in its access shape
// (one wave per trace: read nt floats, write 2 nf + nf + nf + nt floats to four arrays, 16-byte
// accesses, persistent 256 x 512 grid) with no arithmetic.  Its rate is the ceiling the memory
// system offers this traffic pattern; bench.py / DESIGN.md quote it beside the 8 TB/s spec.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

template <bool NT>
__device__ __forceinline__ void probe_store(f4u *p, f4u v)
{
#ifndef THZ_EMU
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
#else
    *p = v;
#endif
}

template <bool NT>
__global__ __launch_bounds__(512) void k_traffic_probe(size_t npix, int nt, int nf, const float *__restrict__ in,
                                                       float *__restrict__ fft, float *__restrict__ amp,
                                                       float *__restrict__ ph, float *__restrict__ out)
{
    const int lane = lane_id();
    const int wpb = (int)(blockDim.x >> 6);
    const size_t stride = (size_t)gridDim.x * wpb;
    for (size_t trace = (size_t)blockIdx.x * wpb + (threadIdx.x >> 6); trace < npix; trace += stride) {
        const float *src = in + trace * (size_t)nt;
        float *f = fft + trace * (size_t)(2 * nf), *a = amp + trace * (size_t)nf, *p = ph + trace * (size_t)nf;
        float *o = out + trace * (size_t)nt;
        // rows of the spectrum arrays start at 8- / 4-byte boundaries: unaligned 16-byte stores,
        // like the epilogue of the real kernel
        for (int e = 4 * lane; e < nt; e += 4 * kWave) {
            const float4 v = *reinterpret_cast<const float4 *>(src + e);
            probe_store<NT>(reinterpret_cast<f4u *>(o + e), f4u{v.x, v.y, v.z, v.w});
            probe_store<NT>(reinterpret_cast<f4u *>(f + e), f4u{v.x, v.y, v.z, v.w});
        }
        for (int e = 4 * lane; e < nt / 2; e += 4 * kWave) {
            const float4 v = *reinterpret_cast<const float4 *>(src + e);
            probe_store<NT>(reinterpret_cast<f4u *>(a + e), f4u{v.x, v.y, v.z, v.w});
            probe_store<NT>(reinterpret_cast<f4u *>(p + e), f4u{v.w, v.z, v.y, v.x});
        }
        if (lane == 0) { f[nt] = 0.f; f[nt + 1] = 0.f; a[nf - 1] = 0.f; p[nf - 1] = 0.f; }
    }
}

void launch_traffic_probe(hipStream_t st, size_t npix, int nt, const float *in, float *fft, float *amp, float *ph,
                          float *out)
{
    // developer knobs for the access-shape experiments of DESIGN.md §6
    const char *eb = getenv("THZ_PROBE_BLOCKS"), *et = getenv("THZ_PROBE_THREADS"), *en = getenv("THZ_PROBE_NT");
    const char *ep = getenv("THZ_PROBE_PITCH");  // row pitch of the spectrum arrays in floats (>= nt/2+1)
    const int blocks = eb ? atoi(eb) : 256, threads = et ? atoi(et) : 512;
    const int nf = ep && atoi(ep) >= nt / 2 + 1 ? atoi(ep) : nt / 2 + 1;
    if (blocks < 1 || blocks > 65536 || threads < 64 || threads > 512 || threads % 64) return;
    if (en && atoi(en))
        THZ_LAUNCH(k_traffic_probe<true>, blocks, threads, 0, st, npix, nt, nf, in, fft, amp, ph, out);
    else
        THZ_LAUNCH(k_traffic_probe<false>, blocks, threads, 0, st, npix, nt, nf, in, fft, amp, ph, out);
}

}  // namespace thz
