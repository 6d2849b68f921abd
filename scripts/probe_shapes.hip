// probe_shapes.hip — measurement aid (developer tool, not part of libthzgpu.so).
//
// Moves the bytes of the fused chain (per trace: read nt floats, write 2 nf + nf + nf + nt floats)
// in DIFFERENT access shapes with no arithmetic, to find out whether the 0.62-0.655 of the 8 TB/s
// roofline that thz_traffic_probe reaches is the memory system's ceiling for a 1 : 3 read / write
// mix or a property of one store pattern.  Pure-read / pure-write / copy
// shapes are timed beside it as reference points.
//
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/probe_shapes scripts/probe_shapes.hip
//   ./gpurun_out/probe_shapes [npix_log2=18]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                    \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f4a __attribute__((ext_vector_type(4)));
constexpr int NT = 4096;

struct Bufs {
    const float *in;
    float *fft, *amp, *ph, *out;
    size_t npix;
    int nf;  // row pitch of amp / ph (fft: 2 nf)
};

template <bool NTS>
__device__ __forceinline__ void st(float *p, f4a v)
{
    if constexpr (NTS) __builtin_nontemporal_store(f4u{v.x, v.y, v.z, v.w}, reinterpret_cast<f4u *>(p));
    else *reinterpret_cast<f4u *>(p) = f4u{v.x, v.y, v.z, v.w};
}

// S0: the shipped probe's shape — loads and stores interleaved element block by element block
template <bool NTS>
__global__ __launch_bounds__(512) void k_interleaved(Bufs B)
{
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const size_t stride = (size_t)gridDim.x * wpb;
    for (size_t t = (size_t)blockIdx.x * wpb + (threadIdx.x >> 6); t < B.npix; t += stride) {
        const float *src = B.in + t * NT;
        float *f = B.fft + t * (size_t)(2 * B.nf), *a = B.amp + t * (size_t)B.nf, *p = B.ph + t * (size_t)B.nf;
        float *o = B.out + t * NT;
        for (int e = 4 * lane; e < NT; e += 256) {
            const f4a v = *reinterpret_cast<const f4a *>(src + e);
            st<NTS>(o + e, v);
            st<NTS>(f + e, v);
        }
        for (int e = 4 * lane; e < NT / 2; e += 256) {
            const f4a v = *reinterpret_cast<const f4a *>(src + e);
            st<NTS>(a + e, v);
            st<NTS>(p + e, v);
        }
    }
}

// S1: the real kernel's phases — whole trace into registers (with the NEXT trace prefetched before
// the stores of this one, like k_f), then spectrum / amp / phase group by group (256 bins each),
// then the time trace
template <bool NTS, bool PREFETCH>
__global__ __launch_bounds__(512) void k_phased(Bufs B)
{
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const size_t stride = (size_t)gridDim.x * wpb;
    size_t t = (size_t)blockIdx.x * wpb + (threadIdx.x >> 6);
    f4a r[16], nx[16];
    if (t < B.npix)
#pragma unroll
        for (int j = 0; j < 16; ++j) nx[j] = *reinterpret_cast<const f4a *>(B.in + t * NT + 4 * (64 * j + lane));
    for (; t < B.npix; t += stride) {
#pragma unroll
        for (int j = 0; j < 16; ++j) r[j] = nx[j];
        if (PREFETCH && t + stride < B.npix) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                nx[j] = *reinterpret_cast<const f4a *>(B.in + (t + stride) * NT + 4 * (64 * j + lane));
        }
        float *f = B.fft + t * (size_t)(2 * B.nf), *a = B.amp + t * (size_t)B.nf, *p = B.ph + t * (size_t)B.nf;
        float *o = B.out + t * NT;
#pragma unroll
        for (int g = 0; g < 8; ++g) {  // 256 bins per group
            st<NTS>(a + 256 * g + 4 * lane, r[g]);
            st<NTS>(f + 512 * g + 8 * lane, r[2 * g]);
            st<NTS>(f + 512 * g + 8 * lane + 4, r[2 * g + 1]);
            st<NTS>(p + 256 * g + 4 * lane, r[g + 8]);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) st<NTS>(o + 4 * (64 * j + lane), r[j]);
        if (!PREFETCH && t + stride < B.npix) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                nx[j] = *reinterpret_cast<const f4a *>(B.in + (t + stride) * NT + 4 * (64 * j + lane));
        }
    }
}

// S2: one output stream — the 48 KiB of a trace's results contiguous (what a single interleaved
// staging buffer per trace would look like)
template <bool NTS>
__global__ __launch_bounds__(512) void k_one_stream(Bufs B)
{
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const size_t stride = (size_t)gridDim.x * wpb;
    size_t t = (size_t)blockIdx.x * wpb + (threadIdx.x >> 6);
    f4a r[16], nx[16];
    if (t < B.npix)
#pragma unroll
        for (int j = 0; j < 16; ++j) nx[j] = *reinterpret_cast<const f4a *>(B.in + t * NT + 4 * (64 * j + lane));
    for (; t < B.npix; t += stride) {
#pragma unroll
        for (int j = 0; j < 16; ++j) r[j] = nx[j];
        if (t + stride < B.npix) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                nx[j] = *reinterpret_cast<const f4a *>(B.in + (t + stride) * NT + 4 * (64 * j + lane));
        }
        float *o = B.fft + t * (size_t)(3 * NT);  // fft buffer is followed by amp, ph, out in one allocation
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int j = 0; j < 16; ++j) st<NTS>(o + q * NT + 4 * (64 * j + lane), r[j]);
    }
}

// S3: writes only (same four arrays, same shape as S1 without the loads)
template <bool NTS>
__global__ __launch_bounds__(512) void k_write_only(Bufs B)
{
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const size_t stride = (size_t)gridDim.x * wpb;
    const f4a v = {1.f, 2.f, 3.f, (float)lane};
    for (size_t t = (size_t)blockIdx.x * wpb + (threadIdx.x >> 6); t < B.npix; t += stride) {
        float *f = B.fft + t * (size_t)(2 * B.nf), *a = B.amp + t * (size_t)B.nf, *p = B.ph + t * (size_t)B.nf;
        float *o = B.out + t * NT;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            st<NTS>(a + 256 * g + 4 * lane, v);
            st<NTS>(f + 512 * g + 8 * lane, v);
            st<NTS>(f + 512 * g + 8 * lane + 4, v);
            st<NTS>(p + 256 * g + 4 * lane, v);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) st<NTS>(o + 4 * (64 * j + lane), v);
    }
}

// S4: reads only (result kept alive through a never-taken store)
__global__ __launch_bounds__(512) void k_read_only(Bufs B)
{
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const size_t stride = (size_t)gridDim.x * wpb;
    f4a acc = {0, 0, 0, 0};
    for (size_t t = (size_t)blockIdx.x * wpb + (threadIdx.x >> 6); t < B.npix; t += stride) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += *reinterpret_cast<const f4a *>(B.in + t * NT + 4 * (64 * j + lane));
    }
    if (acc.x == 12345.678f) B.out[lane] = acc.y + acc.z + acc.w;
}

// S5: copy 1 : 1, wave per trace, persistent
template <bool NTS>
__global__ __launch_bounds__(512) void k_copy_trace(Bufs B)
{
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const size_t stride = (size_t)gridDim.x * wpb;
    for (size_t t = (size_t)blockIdx.x * wpb + (threadIdx.x >> 6); t < B.npix; t += stride) {
        f4a r[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) r[j] = *reinterpret_cast<const f4a *>(B.in + t * NT + 4 * (64 * j + lane));
#pragma unroll
        for (int j = 0; j < 16; ++j) st<NTS>(B.out + t * NT + 4 * (64 * j + lane), r[j]);
    }
}

// S6: classic grid-stride float4 copy (the shape the guide's 6.29 TB/s is quoted on)
__global__ __launch_bounds__(256) void k_copy_flat(const f4a *__restrict__ in, f4a *__restrict__ out, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        out[i] = in[i];
}

// S7: 1 : 3 flat — every float4 read is written to three places of one big output (no trace structure)
__global__ __launch_bounds__(256) void k_flat_1to3(const f4a *__restrict__ in, f4a *__restrict__ out, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f4a v = in[i];
        out[i] = v;
        out[i + n4] = v;
        out[i + 2 * n4] = v;
    }
}

// S8: two kernels' worth in one: block-level phase separation — a block (8 waves) first loads its 8
// traces, barrier, then stores; tests whether bunching reads and writes in time per CU matters
template <bool NTS>
__global__ __launch_bounds__(512) void k_block_phased(Bufs B)
{
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const size_t stride = (size_t)gridDim.x * wpb;
    const size_t t0 = (size_t)blockIdx.x * wpb + (threadIdx.x >> 6);
    const size_t iters = (B.npix + stride - 1) / stride;
    for (size_t it = 0; it < iters; ++it) {
        const size_t t = t0 + it * stride;
        f4a r[16];
        if (t < B.npix)
#pragma unroll
            for (int j = 0; j < 16; ++j) r[j] = *reinterpret_cast<const f4a *>(B.in + t * NT + 4 * (64 * j + lane));
        __syncthreads();
        if (t < B.npix) {
            float *f = B.fft + t * (size_t)(2 * B.nf), *a = B.amp + t * (size_t)B.nf, *p = B.ph + t * (size_t)B.nf;
            float *o = B.out + t * NT;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                st<NTS>(a + 256 * g + 4 * lane, r[g]);
                st<NTS>(f + 512 * g + 8 * lane, r[2 * g]);
                st<NTS>(f + 512 * g + 8 * lane + 4, r[2 * g + 1]);
                st<NTS>(p + 256 * g + 4 * lane, r[g + 8]);
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) st<NTS>(o + 4 * (64 * j + lane), r[j]);
        }
        __syncthreads();
    }
}


// ---- second sweep: what does the store path want? ------------------------------------------
// W1: write only, one contiguous stream, wave per 16 KiB chunk, persistent
__global__ __launch_bounds__(512) void k_w_chunks(float *out, size_t nchunks)
{
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const size_t stride = (size_t)gridDim.x * wpb;
    const f4a v = {1.f, 2.f, 3.f, (float)lane};
    for (size_t t = (size_t)blockIdx.x * wpb + (threadIdx.x >> 6); t < nchunks; t += stride)
#pragma unroll
        for (int j = 0; j < 16; ++j) st<false>(out + t * NT + 4 * (64 * j + lane), v);
}
// W2: write only, flat grid-stride float4
__global__ __launch_bounds__(256) void k_w_flat(f4a *__restrict__ out, size_t n4)
{
    const f4a v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) out[i] = v;
}
// W5: write only, 4 arrays, one array after the other per trace (fft 16 KiB, amp 8, ph 8, out 16)
template <bool BARRIER>
__global__ __launch_bounds__(512) void k_w_seq(Bufs B)
{
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const size_t stride = (size_t)gridDim.x * wpb;
    const f4a v = {1.f, 2.f, 3.f, (float)lane};
    const size_t t0 = (size_t)blockIdx.x * wpb + (threadIdx.x >> 6);
    const size_t iters = (B.npix + stride - 1) / stride;
    for (size_t it = 0; it < iters; ++it) {
        const size_t t = t0 + it * stride;
        if (BARRIER) __syncthreads();
        if (t >= B.npix) continue;
        float *f = B.fft + t * (size_t)(2 * B.nf), *a = B.amp + t * (size_t)B.nf, *p = B.ph + t * (size_t)B.nf;
        float *o = B.out + t * NT;
#pragma unroll
        for (int j = 0; j < 16; ++j) st<false>(f + 4 * (64 * j + lane), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) st<false>(a + 4 * (64 * j + lane), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) st<false>(p + 4 * (64 * j + lane), v);
#pragma unroll
        for (int j = 0; j < 16; ++j) st<false>(o + 4 * (64 * j + lane), v);
    }
}
// W6: write only, block-cooperative: the block's 512 threads sweep the 8 traces' rows of each array as
// ONE contiguous span (8 rows of an array are adjacent in memory): 8 KiB per block-wide instruction
__global__ __launch_bounds__(512) void k_w_coop(Bufs B)
{
    const size_t tpb = 8;
    const f4a v = {1.f, 2.f, 3.f, 4.f};
    const size_t nblk = B.npix / tpb;
    for (size_t b = blockIdx.x; b < nblk; b += gridDim.x) {
        const size_t t = b * tpb;
        float *f = B.fft + t * (size_t)(2 * B.nf), *a = B.amp + t * (size_t)B.nf, *p = B.ph + t * (size_t)B.nf;
        float *o = B.out + t * NT;
        for (int e = 4 * threadIdx.x; e < 8 * NT; e += 2048) st<false>(f + e, v);
        for (int e = 4 * threadIdx.x; e < 4 * NT; e += 2048) st<false>(a + e, v);
        for (int e = 4 * threadIdx.x; e < 4 * NT; e += 2048) st<false>(p + e, v);
        for (int e = 4 * threadIdx.x; e < 8 * NT; e += 2048) st<false>(o + e, v);
    }
}
// S9: like S1 (phased, prefetch) but ONE barrier per trace, right before the store phase: the block's
// eight waves store together, loads run free
template <int NBAR>
__global__ __launch_bounds__(512) void k_phased_bar(Bufs B)
{
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const size_t stride = (size_t)gridDim.x * wpb;
    const size_t t0 = (size_t)blockIdx.x * wpb + (threadIdx.x >> 6);
    const size_t iters = (B.npix + stride - 1) / stride;
    f4a r[16], nx[16];
    if (t0 < B.npix)
#pragma unroll
        for (int j = 0; j < 16; ++j) nx[j] = *reinterpret_cast<const f4a *>(B.in + t0 * NT + 4 * (64 * j + lane));
    for (size_t it = 0; it < iters; ++it) {
        const size_t t = t0 + it * stride;
        const bool on = t < B.npix;
#pragma unroll
        for (int j = 0; j < 16; ++j) r[j] = nx[j];
        if (on && t + stride < B.npix) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                nx[j] = *reinterpret_cast<const f4a *>(B.in + (t + stride) * NT + 4 * (64 * j + lane));
        }
        __syncthreads();
        if (on) {
            float *f = B.fft + t * (size_t)(2 * B.nf), *a = B.amp + t * (size_t)B.nf, *p = B.ph + t * (size_t)B.nf;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                st<false>(a + 256 * g + 4 * lane, r[g]);
                st<false>(f + 512 * g + 8 * lane, r[2 * g]);
                st<false>(f + 512 * g + 8 * lane + 4, r[2 * g + 1]);
                st<false>(p + 256 * g + 4 * lane, r[g + 8]);
            }
        }
        if (NBAR > 1) __syncthreads();
        if (on) {
            float *o = B.out + t * NT;
#pragma unroll
            for (int j = 0; j < 16; ++j) st<false>(o + 4 * (64 * j + lane), r[j]);
        }
    }
}
// S10: S1 with the waves walking CONSECUTIVE traces (wave w owns traces [w c, (w+1) c)): every wave is
// five sequential streams instead of five strided ones
__global__ __launch_bounds__(512) void k_phased_consecutive(Bufs B)
{
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const size_t nw = (size_t)gridDim.x * wpb, w = (size_t)blockIdx.x * wpb + (threadIdx.x >> 6);
    const size_t per = (B.npix + nw - 1) / nw;
    size_t t = w * per;
    const size_t end = (t + per < B.npix) ? t + per : B.npix;
    f4a r[16], nx[16];
    if (t < end)
#pragma unroll
        for (int j = 0; j < 16; ++j) nx[j] = *reinterpret_cast<const f4a *>(B.in + t * NT + 4 * (64 * j + lane));
    for (; t < end; ++t) {
#pragma unroll
        for (int j = 0; j < 16; ++j) r[j] = nx[j];
        if (t + 1 < end) {
#pragma unroll
            for (int j = 0; j < 16; ++j) nx[j] = *reinterpret_cast<const f4a *>(B.in + (t + 1) * NT + 4 * (64 * j + lane));
        }
        float *f = B.fft + t * (size_t)(2 * B.nf), *a = B.amp + t * (size_t)B.nf, *p = B.ph + t * (size_t)B.nf;
        float *o = B.out + t * NT;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            st<false>(a + 256 * g + 4 * lane, r[g]);
            st<false>(f + 512 * g + 8 * lane, r[2 * g]);
            st<false>(f + 512 * g + 8 * lane + 4, r[2 * g + 1]);
            st<false>(p + 256 * g + 4 * lane, r[g + 8]);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) st<false>(o + 4 * (64 * j + lane), r[j]);
    }
}
// S11: block-cooperative stores behind per-wave loads: the eight traces' values go through LDS and the
// block writes each array's 8 adjacent rows as one contiguous span (stands for "stage the outputs in
// LDS, store block-wide")
__global__ __launch_bounds__(512) void k_coop_store(Bufs B)
{
    extern __shared__ __align__(16) float lds[];  // 8 x 4096 floats
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t nblk = B.npix / 8;
    for (size_t b = blockIdx.x; b < nblk; b += gridDim.x) {
        const size_t t = b * 8;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            *reinterpret_cast<f4a *>(lds + wv * NT + 4 * (64 * j + lane)) =
                *reinterpret_cast<const f4a *>(B.in + (t + wv) * NT + 4 * (64 * j + lane));
        __syncthreads();
        float *f = B.fft + t * (size_t)(2 * B.nf), *a = B.amp + t * (size_t)B.nf, *p = B.ph + t * (size_t)B.nf;
        float *o = B.out + t * NT;
        for (int e = 4 * threadIdx.x; e < 8 * NT; e += 2048) st<false>(f + e, *reinterpret_cast<f4a *>(lds + e));
        for (int e = 4 * threadIdx.x; e < 4 * NT; e += 2048) st<false>(a + e, *reinterpret_cast<f4a *>(lds + e));
        for (int e = 4 * threadIdx.x; e < 4 * NT; e += 2048) st<false>(p + e, *reinterpret_cast<f4a *>(lds + 4 * NT + e));
        for (int e = 4 * threadIdx.x; e < 8 * NT; e += 2048) st<false>(o + e, *reinterpret_cast<f4a *>(lds + e));
        __syncthreads();
    }
}

template <class F>
static double time_ms(hipStream_t s, F launch, int reps = 7)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    launch();
    CK(hipStreamSynchronize(s));
    std::vector<float> ms(reps);
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(a, s));
        launch();
        CK(hipEventRecord(b, s));
        CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms[i], a, b));
    }
    std::sort(ms.begin(), ms.end());
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return ms[reps / 2];
}

int main(int argc, char **argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 18;
    const size_t npix = (size_t)1 << lg;
    const int nf = NT / 2 + 1;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float *in, *big;
    const size_t in_f = npix * NT;
    // one allocation for the outputs so that S2 can treat it as a single stream: fft | amp | ph | out
    const size_t nfp = NT / 2 + 4;  // padded pitch variant needs the room
    const size_t big_f = npix * (2 * nfp + nfp + nfp + NT) + 64;
    CK(hipMalloc((void **)&in, in_f * 4));
    CK(hipMalloc((void **)&big, big_f * 4));
    CK(hipMemsetAsync(in, 0, in_f * 4, s));
    CK(hipMemsetAsync(big, 0, big_f * 4, s));
    CK(hipStreamSynchronize(s));
    auto bufs = [&](int pitch) {
        Bufs B;
        B.in = in; B.npix = npix; B.nf = pitch;
        B.fft = big;
        B.amp = B.fft + npix * 2 * (size_t)pitch;
        B.ph = B.amp + npix * (size_t)pitch;
        B.out = B.ph + npix * (size_t)pitch;
        return B;
    };
    const double m_full = (double)npix * (4.0 * NT + 16.0 * nf + 4.0 * NT);  // algorithmic bytes (dense layout)
    const double rd = (double)npix * 4.0 * NT;
    auto report = [&](const char *name, double ms, double bytes) {
        printf("%-64s %8.3f ms  %7.1f GB/s  %5.3f of 8 TB/s\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
        fflush(stdout);
    };
    printf("npix = %zu traces of %d samples; M_full bytes = %.2f GB\n", npix, NT, m_full / 1e9);
    Bufs D = bufs(nf), Pd = bufs(NT / 2 + 4);

    if (argc > 2 && atoi(argv[2]) == 2) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_coop_store), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * NT * 4));
        const double wr = m_full - rd;
        report("S1 phased like k_f (prefetch), 256 x 512  [baseline]", time_ms(s, [&] { hipLaunchKernelGGL((k_phased<false, true>), dim3(256), dim3(512), 0, s, D); }), m_full);
        report("memset 48 KiB/trace", time_ms(s, [&] { CK(hipMemsetAsync(big, 0, npix * 12ull * NT, s)); }), npix * 12.0 * NT);
        for (int blocks : {256, 1024}) {
            char nm[128];
            snprintf(nm, sizeof nm, "W1 write only, one stream, wave per 16 KiB chunk, %d x 512", blocks);
            report(nm, time_ms(s, [&] { hipLaunchKernelGGL(k_w_chunks, dim3(blocks), dim3(512), 0, s, big, npix * 3); }), npix * 12.0 * NT);
        }
        for (int blocks : {2048, 16384, 131072}) {
            char nm[128];
            snprintf(nm, sizeof nm, "W2 write only, flat float4 grid-stride, %d x 256", blocks);
            report(nm, time_ms(s, [&] { hipLaunchKernelGGL(k_w_flat, dim3(blocks), dim3(256), 0, s, (f4a *)big, npix * 3 * (size_t)(NT / 4)); }), npix * 12.0 * NT);
        }
        report("S3 write only 4 arrays (a,f,f,p per group, then out), 256 x 512", time_ms(s, [&] { hipLaunchKernelGGL(k_write_only<false>, dim3(256), dim3(512), 0, s, D); }), wr);
        report("S3 same, padded rows", time_ms(s, [&] { hipLaunchKernelGGL(k_write_only<false>, dim3(256), dim3(512), 0, s, Pd); }), wr);
        report("S3 same, 256 x 256 (4 waves)", time_ms(s, [&] { hipLaunchKernelGGL(k_write_only<false>, dim3(256), dim3(256), 0, s, D); }), wr);
        report("S3 same, 256 x 128 (2 waves)", time_ms(s, [&] { hipLaunchKernelGGL(k_write_only<false>, dim3(256), dim3(128), 0, s, D); }), wr);
        report("W5 write only, array after array, 256 x 512", time_ms(s, [&] { hipLaunchKernelGGL(k_w_seq<false>, dim3(256), dim3(512), 0, s, D); }), wr);
        report("W5 same with a block barrier per trace", time_ms(s, [&] { hipLaunchKernelGGL(k_w_seq<true>, dim3(256), dim3(512), 0, s, D); }), wr);
        report("W6 write only, block-cooperative spans, 256 x 512", time_ms(s, [&] { hipLaunchKernelGGL(k_w_coop, dim3(256), dim3(512), 0, s, D); }), wr);
        report("W6 same, 512 x 512", time_ms(s, [&] { hipLaunchKernelGGL(k_w_coop, dim3(512), dim3(512), 0, s, D); }), wr);
        report("W6 same, 1024 x 512", time_ms(s, [&] { hipLaunchKernelGGL(k_w_coop, dim3(1024), dim3(512), 0, s, D); }), wr);
        report("S8 block-phased (2 barriers: loads | stores)", time_ms(s, [&] { hipLaunchKernelGGL(k_block_phased<false>, dim3(256), dim3(512), 0, s, D); }), m_full);
        report("S9 prefetch + ONE barrier before the stores", time_ms(s, [&] { hipLaunchKernelGGL(k_phased_bar<1>, dim3(256), dim3(512), 0, s, D); }), m_full);
        report("S9 prefetch + barriers before spectrum and before time stores", time_ms(s, [&] { hipLaunchKernelGGL(k_phased_bar<2>, dim3(256), dim3(512), 0, s, D); }), m_full);
        report("S9 one barrier, 256 x 448 (7 waves)", time_ms(s, [&] { hipLaunchKernelGGL(k_phased_bar<1>, dim3(256), dim3(448), 0, s, D); }), m_full);
        report("S10 waves walk consecutive traces, 256 x 512", time_ms(s, [&] { hipLaunchKernelGGL(k_phased_consecutive, dim3(256), dim3(512), 0, s, D); }), m_full);
        report("S11 loads -> LDS -> block-cooperative contiguous stores, 256 x 512", time_ms(s, [&] { hipLaunchKernelGGL(k_coop_store, dim3(256), dim3(512), 8 * NT * 4, s, D); }), m_full);
        report("S11 same, 512 x 512", time_ms(s, [&] { hipLaunchKernelGGL(k_coop_store, dim3(512), dim3(512), 8 * NT * 4, s, D); }), m_full);
        report("S1 phased like k_f (prefetch), 256 x 512  [baseline again]", time_ms(s, [&] { hipLaunchKernelGGL((k_phased<false, true>), dim3(256), dim3(512), 0, s, D); }), m_full);
        CK(hipFree(in));
        CK(hipFree(big));
        return 0;
    }
    report("memset 48 KiB/trace (hipMemsetAsync)", time_ms(s, [&] { CK(hipMemsetAsync(big, 0, npix * 12ull * NT, s)); }), npix * 12.0 * NT);
    report("hipMemcpyDtoD 16 KiB/trace", time_ms(s, [&] { CK(hipMemcpyAsync(big, in, in_f * 4, hipMemcpyDeviceToDevice, s)); }), 2 * rd);
    for (int blocks : {256, 512, 1024, 4096}) {
        char nm[128];
        snprintf(nm, sizeof nm, "S6 flat float4 copy, %d x 256", blocks * 8);
        report(nm, time_ms(s, [&] { hipLaunchKernelGGL(k_copy_flat, dim3(blocks * 8), dim3(256), 0, s, (const f4a *)in, (f4a *)big, in_f / 4); }), 2 * rd);
    }
    report("S7 flat 1:3 (one read, three writes), 8192 x 256", time_ms(s, [&] { hipLaunchKernelGGL(k_flat_1to3, dim3(8192), dim3(256), 0, s, (const f4a *)in, (f4a *)big, in_f / 4); }), 4 * rd);
    report("S4 read only, wave per trace, 256 x 512", time_ms(s, [&] { hipLaunchKernelGGL(k_read_only, dim3(256), dim3(512), 0, s, D); }), rd);
    report("S4 read only, wave per trace, 1024 x 512", time_ms(s, [&] { hipLaunchKernelGGL(k_read_only, dim3(1024), dim3(512), 0, s, D); }), rd);
    report("S3 write only 4 arrays, 256 x 512", time_ms(s, [&] { hipLaunchKernelGGL(k_write_only<false>, dim3(256), dim3(512), 0, s, D); }), m_full - rd);
    report("S3 write only 4 arrays, nontemporal", time_ms(s, [&] { hipLaunchKernelGGL(k_write_only<true>, dim3(256), dim3(512), 0, s, D); }), m_full - rd);
    report("S3 write only 4 arrays, 1024 x 512", time_ms(s, [&] { hipLaunchKernelGGL(k_write_only<false>, dim3(1024), dim3(512), 0, s, D); }), m_full - rd);
    report("S5 copy wave per trace, 256 x 512", time_ms(s, [&] { hipLaunchKernelGGL(k_copy_trace<false>, dim3(256), dim3(512), 0, s, D); }), 2 * rd);
    report("S5 copy wave per trace, 1024 x 512", time_ms(s, [&] { hipLaunchKernelGGL(k_copy_trace<false>, dim3(1024), dim3(512), 0, s, D); }), 2 * rd);
    report("S5 copy wave per trace, nontemporal, 256 x 512", time_ms(s, [&] { hipLaunchKernelGGL(k_copy_trace<true>, dim3(256), dim3(512), 0, s, D); }), 2 * rd);
    for (int blocks : {256, 512, 1024}) {
        char nm[128];
        snprintf(nm, sizeof nm, "S0 interleaved (shipped probe), %d x 512", blocks);
        report(nm, time_ms(s, [&] { hipLaunchKernelGGL(k_interleaved<false>, dim3(blocks), dim3(512), 0, s, D); }), m_full);
    }
    report("S0 interleaved, nontemporal, 256 x 512", time_ms(s, [&] { hipLaunchKernelGGL(k_interleaved<true>, dim3(256), dim3(512), 0, s, D); }), m_full);
    for (int blocks : {256, 512, 1024}) {
        char nm[128];
        snprintf(nm, sizeof nm, "S1 phased like k_f (prefetch), %d x 512", blocks);
        report(nm, time_ms(s, [&] { hipLaunchKernelGGL((k_phased<false, true>), dim3(blocks), dim3(512), 0, s, D); }), m_full);
    }
    report("S1 phased, prefetch, nontemporal, 256 x 512", time_ms(s, [&] { hipLaunchKernelGGL((k_phased<true, true>), dim3(256), dim3(512), 0, s, D); }), m_full);
    report("S1 phased, load after stores, 256 x 512", time_ms(s, [&] { hipLaunchKernelGGL((k_phased<false, false>), dim3(256), dim3(512), 0, s, D); }), m_full);
    report("S1 phased, prefetch, 256 x 448 (7 waves)", time_ms(s, [&] { hipLaunchKernelGGL((k_phased<false, true>), dim3(256), dim3(448), 0, s, D); }), m_full);
    report("S1 phased, prefetch, 256 x 384 (6 waves)", time_ms(s, [&] { hipLaunchKernelGGL((k_phased<false, true>), dim3(256), dim3(384), 0, s, D); }), m_full);
    report("S1 phased, prefetch, 256 x 256 (4 waves)", time_ms(s, [&] { hipLaunchKernelGGL((k_phased<false, true>), dim3(256), dim3(256), 0, s, D); }), m_full);
    report("S1 phased, prefetch, padded rows (nf+3), 256 x 512", time_ms(s, [&] { hipLaunchKernelGGL((k_phased<false, true>), dim3(256), dim3(512), 0, s, Pd); }), m_full);
    report("S2 one 48 KiB output stream, 256 x 512", time_ms(s, [&] { hipLaunchKernelGGL(k_one_stream<false>, dim3(256), dim3(512), 0, s, D); }), 4 * rd);
    report("S2 one 48 KiB output stream, nontemporal", time_ms(s, [&] { hipLaunchKernelGGL(k_one_stream<true>, dim3(256), dim3(512), 0, s, D); }), 4 * rd);
    report("S8 block-phased (barrier between loads and stores)", time_ms(s, [&] { hipLaunchKernelGGL(k_block_phased<false>, dim3(256), dim3(512), 0, s, D); }), m_full);
    report("S8 block-phased, 512 x 512", time_ms(s, [&] { hipLaunchKernelGGL(k_block_phased<false>, dim3(512), dim3(512), 0, s, D); }), m_full);
    CK(hipFree(in));
    CK(hipFree(big));
    return 0;
}
