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

// ---- cross-lane moves on the DPP path (no LDS crossbar traffic, unlike __shfl)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_or_zero(float v)
{
    // lanes whose source is out of range, or whose row is masked off, read 0
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
// value of lane-1 (0 in lane 0)
__device__ __forceinline__ float wave_shr1(float v) { return dpp_or_zero<0x138, 0xf>(v); }
// value of a fixed lane, in every lane (wave-uniform result)
template <int LANE>
__device__ __forceinline__ float wave_bcast(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), LANE));
}
// inclusive prefix sum over the 64 lanes: 4 row_shr steps inside each row of 16,
// then row_bcast:15 / row_bcast:31 (the GFX9 wave64 scan)
__device__ __forceinline__ float wave_scan_add(float v)
{
    v += dpp_or_zero<0x111, 0xf>(v);  // row_shr:1
    v += dpp_or_zero<0x112, 0xf>(v);  // row_shr:2
    v += dpp_or_zero<0x114, 0xf>(v);  // row_shr:4
    v += dpp_or_zero<0x118, 0xf>(v);  // row_shr:8
    v += dpp_or_zero<0x142, 0xa>(v);  // row_bcast:15 -> rows 1, 3
    v += dpp_or_zero<0x143, 0xc>(v);  // row_bcast:31 -> rows 2, 3
    return v;
}
// sum over the 64 lanes, result in every lane
__device__ __forceinline__ float wave_reduce_add(float v) { return wave_bcast<kWave - 1>(wave_scan_add(v)); }

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
    // p + (opaque 0), not an opaque p: the pointer keeps its provenance, so the compiler still
    // knows it is global (kernel argument) or LDS and emits global_load / ds_read.  An opaque
    // pointer turns every access into flat_load, and a flat result can only be waited for with
    // vmcnt(0) lgkmcnt(0) — which also waits for every store the wave has in flight.
    int z = 0;
    asm volatile("" : "+s"(z));
    return p + z;
}

// Same for a per-lane integer: makes a lane-dependent base index opaque, so that
// (a) every access becomes base + immediate offset and (b) the optimiser cannot
// pre-compute (and keep live, or spill) one address VGPR per unrolled access.
__device__ __forceinline__ int launder_v(int x)
{
    asm volatile("" : "+v"(x));
    return x;
}

// x, but not computable before `dep` is: serialises batches of independent loads that the
// scheduler would otherwise all issue up front (and spill their destinations)
__device__ __forceinline__ int launder_after(int x, float dep)
{
    asm volatile("" : "+v"(x) : "v"(dep));
    return x;
}

// a float the optimiser must have in a register here (forces pending loads of it to complete)
__device__ __forceinline__ float launder_f(float x)
{
    asm volatile("" : "+v"(x));
    return x;
}

// value known to be the same in every lane of the wave -> keep it in an SGPR
#define THZ_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)

#define THZ_LAUNCH(kernel, grid, block, lds_bytes, stream, ...) \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (lds_bytes), (stream), __VA_ARGS__)

#endif  // !THZ_EMU

// ---- lean single-precision elementary functions for the spectrum epilogue.
// OCML's atan2f/sqrtf are IEEE-careful (denormal scaling, inf/nan lattice:
// ~60 and ~12 instructions); the epilogue evaluates one of each per bin, which
// made it cost more VALU issue than the transform itself.
#ifdef THZ_EMU
__device__ __forceinline__ float fast_rcp(float x) { return 1.0f / x; }
__device__ __forceinline__ float fast_sqrt(float x) { return sqrtf(x); }
#else
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }   // 1 ulp
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); } // 1 ulp
#endif

// atan2f to ~2e-7 rad: octant reduction to a = min/max in [0,1], the Cephes atanf
// second reduction t = (a-1)/(a+1) for a > tan(pi/8), degree-4 polynomial in t^2.
// atan2(0, 0) = 0 (num_complex::arg's atan2 gives 0 or pi depending on the
// sign of zero; bins that are exactly zero carry no phase information).
__device__ __forceinline__ float fast_atan2f(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    float a = mn * fast_rcp(mx);
    if (!(mx > 0.0f)) a = 0.0f;  // 0/0
    const bool hi = a > 0.41421356237309503f;
    const float t = hi ? (a - 1.0f) * fast_rcp(a + 1.0f) : a;
    const float z = t * t;
    float r = fmaf(fmaf(fmaf(fmaf(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z,
                        -3.33329491539e-1f) * z,
                   t, t);
    r += hi ? 0.78539816339744831f : 0.0f;
    r = (ay > ax) ? 1.57079632679489662f - r : r;
    r = (x < 0.0f) ? 3.14159265358979324f - r : r;
    return copysignf(r, y);
}

struct alignas(8) c32 {
    float re, im;
};
// two adjacent complex values: the unit of every 16-byte LDS access
struct alignas(16) c32x2 {
    c32 a, b;
};

__device__ __forceinline__ c32 cmul(c32 a, c32 b)
{
    return c32{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
__device__ __forceinline__ c32 cadd(c32 a, c32 b) { return c32{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ c32 csub(c32 a, c32 b) { return c32{a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ c32 cconj(c32 a) { return c32{a.re, -a.im}; }

}  // namespace thz
