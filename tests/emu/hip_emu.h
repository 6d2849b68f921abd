// hip_emu.h — minimal host emulation of the HIP constructs the thz kernels use.
//
// TEST INFRASTRUCTURE ONLY.  tests/emu/build_emu.py compiles the product's
// csrc/kernels.hip with -DTHZ_EMU against this header so that the kernels'
// index arithmetic (LDS exchanges, lane->bin maps, scans) can be checked on a
// CPU-only container.  Every lane is a host thread; blocks run one after the
// other; wave_sync()/__syncthreads() are pthread barriers.  It is slow and it
// is never linked into libthzgpu.so.
#pragma once

#include <pthread.h>
#include <sched.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __align__(n) __attribute__((aligned(n)))

typedef void *hipStream_t;

struct float2 {
    float x, y;
};
struct float4 {
    float x, y, z, w;
};
struct uint4 {
    unsigned int x, y, z, w;
};
static inline uint4 make_uint4(unsigned int x, unsigned int y, unsigned int z, unsigned int w) { return uint4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

namespace thz_emu {

struct Dim3 {
    unsigned x = 1, y = 1, z = 1;
};

inline thread_local Dim3 t_threadIdx, t_blockIdx;
inline Dim3 g_blockDim, g_gridDim;
inline unsigned char *g_dyn_lds = nullptr;

struct BlockSync {
    pthread_barrier_t block;
    std::vector<pthread_barrier_t> waves;
    std::vector<float> slots;
};
inline BlockSync *g_sync = nullptr;

template <class F>
void run_grid(unsigned grid, unsigned block, size_t lds_bytes, F body)
{
    if (block % 64 != 0) {
        std::fprintf(stderr, "hip_emu: block size must be a multiple of 64\n");
        std::abort();
    }
    g_blockDim.x = block;
    g_gridDim.x = grid;
    const unsigned nwaves = block / 64;
    for (unsigned b = 0; b < grid; ++b) {
        BlockSync sync;
        pthread_barrier_init(&sync.block, nullptr, block);
        sync.waves.resize(nwaves);
        for (auto &w : sync.waves) pthread_barrier_init(&w, nullptr, 64);
        sync.slots.assign((size_t)block, 0.f);
        g_sync = &sync;
        void *raw = nullptr;
        if (posix_memalign(&raw, 64, lds_bytes + 64) != 0) std::abort();
        g_dyn_lds = (unsigned char *)raw;
        std::memset(g_dyn_lds, 0xCD, lds_bytes + 64);
        std::vector<std::thread> th;
        th.reserve(block);
        for (unsigned t = 0; t < block; ++t) {
            th.emplace_back([&, t, b]() {
                t_threadIdx.x = t;
                t_blockIdx.x = b;
                body();
            });
        }
        for (auto &x : th) x.join();
        std::free(raw);
        g_dyn_lds = nullptr;
        for (auto &w : sync.waves) pthread_barrier_destroy(&w);
        pthread_barrier_destroy(&sync.block);
        g_sync = nullptr;
    }
}

}  // namespace thz_emu

#define threadIdx thz_emu::t_threadIdx
#define blockIdx thz_emu::t_blockIdx
#define blockDim thz_emu::g_blockDim
#define gridDim thz_emu::g_gridDim

static inline unsigned int atomicOr(unsigned int *p, unsigned int v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline void __syncthreads() { pthread_barrier_wait(&thz_emu::g_sync->block); }

namespace thz {

inline void wave_sync() { pthread_barrier_wait(&thz_emu::g_sync->waves[threadIdx.x / 64]); }
inline void block_lds_barrier() { __syncthreads(); }
inline unsigned lds_flag_load(const unsigned *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
inline void lds_flag_store(unsigned *p, unsigned v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
inline void spin_pause() { sched_yield(); }
typedef int thz_i8 __attribute__((ext_vector_type(8), aligned(16)));   // 16-byte alignment: what a record has
typedef int thz_i16 __attribute__((ext_vector_type(16), aligned(16)));
inline void want_scalars_now(thz_i16, thz_i8, int) {}
inline int lane_id() { return (int)(threadIdx.x & 63); }

inline float wave_shfl(float v, int src)
{
    float *s = thz_emu::g_sync->slots.data() + (threadIdx.x / 64) * 64;
    s[threadIdx.x & 63] = v;
    wave_sync();
    float r = s[src & 63];
    wave_sync();
    return r;
}
inline void row_bcast16(float v, float (&out)[16])
{
    float *s = thz_emu::g_sync->slots.data() + (threadIdx.x / 64) * 64;
    s[threadIdx.x & 63] = v;
    wave_sync();
    for (int q = 0; q < 16; ++q) out[q] = s[(threadIdx.x & 48) + q];
    wave_sync();
}
inline void row_mul16(const float (&a)[16], float k, float (&prod)[16])
{
    float kv[16];
    row_bcast16(k, kv);
    for (int q = 0; q < 16; ++q) prod[q] = a[q] * kv[q];
}
inline float wave_shfl_up(float v, int d)
{
    int l = lane_id();
    float r = wave_shfl(v, l - d >= 0 ? l - d : l);
    return r;
}
inline float wave_shfl_xor(float v, int m) { return wave_shfl(v, lane_id() ^ m); }
inline float wave_shr1(float v)
{
    float r = wave_shfl(v, lane_id() > 0 ? lane_id() - 1 : 0);
    return lane_id() > 0 ? r : 0.0f;
}
template <int LANE>
inline float wave_bcast(float v) { return wave_shfl(v, LANE); }
inline float wave_scan_add(float v)
{
    const int l = lane_id();
    for (int d = 1; d < 64; d <<= 1) {
        float t = wave_shfl_up(v, d);
        if (l >= d) v += t;
    }
    return v;
}
inline float wave_reduce_add(float v) { return wave_bcast<63>(wave_scan_add(v)); }

}  // namespace thz

#define THZ_DYN_LDS(name) unsigned char *name = thz_emu::g_dyn_lds
#define THZ_WAVES_PER_SIMD(n)
namespace thz {
template <class T>
inline const T *launder_uniform(const T *p) { return p; }
inline int launder_v(int x) { return x; }
inline float launder_f(float x) { return x; }
inline int launder_after(int x, float) { return x; }
}  // namespace thz
#define THZ_UNIFORM(x) (x)
#define THZ_SCHED_FENCE() ((void)0)

#define THZ_LAUNCH(kernel, grid, block, lds_bytes, stream, ...) \
    thz_emu::run_grid((unsigned)(grid), (unsigned)(block), (size_t)(lds_bytes), [&]() { kernel(__VA_ARGS__); })
