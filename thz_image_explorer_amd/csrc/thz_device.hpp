// thz_device.hpp — the few device-side primitives every kernel uses.
//
// The kernels are written once, for gfx950 (wave64).  When THZ_EMU is defined
// the same source is compiled by the host clang++ against tests/emu/hip_emu.h,
// which runs every lane as a host thread: that build exists only so that the
// index arithmetic of the kernels can be unit-tested in a container that has
// no GPU.  It is test infrastructure, never shipped and never a fallback —
// libthzgpu.so is always built without THZ_EMU.
#pragma once

#ifdef THZ_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#endif

#include <stdint.h>

namespace thz {

constexpr int kWave = 64;  // gfx950 wavefront

#ifndef THZ_EMU

// Orders this wave's LDS traffic: everything written to LDS by any lane
// before the call is visible to every lane after it.  DS instructions of one
// wave execute in issue order, so no s_barrier is needed — only a compiler
// fence so that loads are not hoisted above the stores of other lanes.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (kWave - 1)); }

__device__ __forceinline__ float wave_shfl(float v, int src) { return __shfl(v, src, kWave); }
__device__ __forceinline__ float wave_shfl_up(float v, int d) { return __shfl_up(v, d, kWave); }
__device__ __forceinline__ float wave_shfl_xor(float v, int m) { return __shfl_xor(v, m, kWave); }

#define THZ_DYN_LDS(name) extern __shared__ __align__(16) unsigned char name[]

// Stops the machine scheduler from moving instructions across this point: the
// fully unrolled passes otherwise get all their LDS/global loads hoisted to the
// top, which blows the VGPR budget of 256 (2 waves per SIMD) and spills.
#define THZ_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// Hides a (wave-uniform) pointer's value from the optimiser for one loop
// iteration, so that loads of trace-invariant tables (windows) stay inside the
// trace loop instead of being hoisted into dozens of permanently live VGPRs.
template <class T>
__device__ __forceinline__ const T *launder_uniform(const T *p)
{
    asm volatile("" : "+s"(p));
    return p;
}

// Same for a per-lane integer: makes a lane-dependent base index opaque, so that
// (a) every access becomes base + immediate offset and (b) the optimiser cannot
// pre-compute (and keep live, or spill) one address VGPR per unrolled access.
__device__ __forceinline__ int launder_v(int x)
{
    asm volatile("" : "+v"(x));
    return x;
}

// value known to be the same in every lane of the wave -> keep it in an SGPR
#define THZ_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)

#define THZ_LAUNCH(kernel, grid, block, lds_bytes, stream, ...) \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (lds_bytes), (stream), __VA_ARGS__)

#endif  // !THZ_EMU

// sum over the 64 lanes, result in every lane
__device__ __forceinline__ float wave_reduce_add(float v)
{
#pragma unroll
    for (int m = kWave / 2; m >= 1; m >>= 1) v += wave_shfl_xor(v, m);
    return v;
}

// inclusive prefix sum over lanes
__device__ __forceinline__ float wave_scan_add(float v)
{
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        float t = wave_shfl_up(v, d);
        if (l >= d) v += t;
    }
    return v;
}

struct c32 {
    float re, im;
};

__device__ __forceinline__ c32 cmul(c32 a, c32 b)
{
    return c32{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
__device__ __forceinline__ c32 cadd(c32 a, c32 b) { return c32{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ c32 csub(c32 a, c32 b) { return c32{a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ c32 cconj(c32 a) { return c32{a.re, -a.im}; }

}  // namespace thz
