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

// Block barrier that orders LDS traffic only: this wave's LDS writes are complete before it signals, and no LDS
// read is moved above the barrier — but global stores stay in flight (`__syncthreads()` is a full workgroup fence:
// it would drain every outstanding store of the wave at each call).
__device__ __forceinline__ void block_lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// A flag in LDS handed from wave to wave of a block: relaxed atomic accesses plus LDS-scoped fences, so that — like
// block_lds_barrier — neither side waits for its global stores.
__device__ __forceinline__ unsigned lds_flag_load(const unsigned *p)
{
    const unsigned v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    return v;
}
__device__ __forceinline__ void lds_flag_store(unsigned *p, unsigned v)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void spin_pause() { __builtin_amdgcn_s_sleep(1); }
// Block-uniform values the compiler must have in scalar registers HERE, all of them at once.  Used on a per-block
// record in front of the block's first branch: left alone, the compiler loads such a record piece by piece, each
// piece behind the branch that first needs it — five dependent scalar round trips at the head of a 10 us kernel.
typedef int thz_i8 __attribute__((ext_vector_type(8), aligned(16)));   // 16-byte alignment: what a record has
typedef int thz_i16 __attribute__((ext_vector_type(16), aligned(16)));
__device__ __forceinline__ void want_scalars_now(thz_i16 a, thz_i8 b, int c) { asm volatile("" ::"s"(a), "s"(b), "s"(c)); }

__device__ __forceinline__ float wave_shfl(float v, int src) { return __shfl(v, src, kWave); }
__device__ __forceinline__ float wave_shfl_up(float v, int d) { return __shfl_up(v, d, kWave); }
__device__ __forceinline__ float wave_shfl_xor(float v, int m) { return __shfl_xor(v, m, kWave); }

// out[q] = v of lane q of the caller's row of 16 lanes, q = 0..15 (DPP row_newbcast: each folds into the VALU
// instruction that consumes it, so a value held one per lane is handed to a whole row at no cost).  All 16 lanes of
// the row must be active.
template <int Q>
__device__ __forceinline__ float row_lane(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + Q, 0xf, 0xf, true));
}
__device__ __forceinline__ void row_bcast16(float v, float (&out)[16])
{
    out[0] = row_lane<0>(v); out[1] = row_lane<1>(v); out[2] = row_lane<2>(v); out[3] = row_lane<3>(v);
    out[4] = row_lane<4>(v); out[5] = row_lane<5>(v); out[6] = row_lane<6>(v); out[7] = row_lane<7>(v);
    out[8] = row_lane<8>(v); out[9] = row_lane<9>(v); out[10] = row_lane<10>(v); out[11] = row_lane<11>(v);
    out[12] = row_lane<12>(v); out[13] = row_lane<13>(v); out[14] = row_lane<14>(v); out[15] = row_lane<15>(v);
}

// prod[q] = a[q] * (k of lane q of the caller's row of 16 lanes): sixteen v_mul_f32_dpp.  Written out because the
// compiler, left to fold row_lane<q>() into the multiplies, pairs the multiplies into v_pk_mul_f32 first (which takes
// no DPP operand) and keeps most broadcasts as separate moves.  k must come from memory or LDS (a VALU result read
// by DPP needs two wait states, which inline assembly does not get inserted for it; the s_nop covers a copy).
__device__ __forceinline__ void row_mul16(const float (&a)[16], float k, float (&prod)[16])
{
#define THZ_DPP_MUL(d, s, q) "v_mul_f32_dpp " d ", %8, " s " row_newbcast:" #q " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
    asm("s_nop 1\n" THZ_DPP_MUL("%0", "%9", 0) THZ_DPP_MUL("%1", "%10", 1) THZ_DPP_MUL("%2", "%11", 2) THZ_DPP_MUL("%3", "%12", 3)
            THZ_DPP_MUL("%4", "%13", 4) THZ_DPP_MUL("%5", "%14", 5) THZ_DPP_MUL("%6", "%15", 6) THZ_DPP_MUL("%7", "%16", 7)
        : "=&v"(prod[0]), "=&v"(prod[1]), "=&v"(prod[2]), "=&v"(prod[3]), "=&v"(prod[4]), "=&v"(prod[5]), "=&v"(prod[6]), "=&v"(prod[7])
        : "v"(k), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
    asm("s_nop 1\n" THZ_DPP_MUL("%0", "%9", 8) THZ_DPP_MUL("%1", "%10", 9) THZ_DPP_MUL("%2", "%11", 10) THZ_DPP_MUL("%3", "%12", 11)
            THZ_DPP_MUL("%4", "%13", 12) THZ_DPP_MUL("%5", "%14", 13) THZ_DPP_MUL("%6", "%15", 14) THZ_DPP_MUL("%7", "%16", 15)
        : "=&v"(prod[8]), "=&v"(prod[9]), "=&v"(prod[10]), "=&v"(prod[11]), "=&v"(prod[12]), "=&v"(prod[13]), "=&v"(prod[14]), "=&v"(prod[15])
        : "v"(k), "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]), "v"(a[15]));
#undef THZ_DPP_MUL
}

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
// waves per SIMD a kernel is compiled for (register budget 512 / n); nothing in the emulation
#define THZ_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, n)))

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

// ---- x / d for a loop-invariant divisor (the inverse transform's 1 / nt when nt is not a power of two, where
// the compiler cannot turn the division into an exact multiplication): with y = RN(1 / d),
//     q = RN(x y),   r = x - q d  (exact inside an FMA),   q' = RN(q + r y)
// q' is the correctly rounded quotient (Markstein 1990; no mismatch against `x / d` in 1.4e9 random operands
// for d = 777 ... 8191) in three instructions instead of the eleven of the generic IEEE sequence (v_div_scale x2,
// v_rcp, four FMAs, v_div_fmas, v_div_fixup) — which was 14 % of the nt = 1001 chain's instruction issue.
// PRECONDITION: finite x whose quotient is a normal number (what samples of a scan divided by a trace length are).
// x = +-Inf gives NaN where IEEE division gives +-Inf (r = Inf - Inf), -0 comes back as +0, and a quotient in the
// denormal / flushed range can be one ulp off; the emulation test pins both cases (test_emu_kernels.py::test_div_const_edge_cases).
struct DivConst {
    float d, rcp;
    __device__ __forceinline__ explicit DivConst(float divisor) : d(divisor), rcp(1.0f / divisor) {}
    __device__ __forceinline__ float operator()(float x) const
    {
        const float q = x * rcp;
        const float r = __builtin_fmaf(-q, d, x);
        return __builtin_fmaf(r, rcp, q);
    }
};

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

// atan2f to 3e-7 rad: octant reduction to a = min/max in [0,1], then ONE odd polynomial over the
// whole octant, atan(a) = a + a^3 Q(a^2) with Q of degree 7 (Chebyshev-node fit of
// (atan(sqrt z)/sqrt z - 1)/z on [0,1]: 5.5e-8 in exact arithmetic; the rest of the 3e-7 is the ratio's
// 1.5 ulp and the half ulp of pi/2 - r and pi - r, 1.2e-7 at |angle| ~ pi).  The Cephes form it replaces
// had a second range reduction t = (a-1)/(a+1) — a compare, two adds, two selects and a second
// reciprocal, which on CDNA costs four issue slots by itself — in front of a degree-4 polynomial; four
// more FMAs are cheaper, and the polynomial of four bins runs as packed math (fast_atan2f_x4).
// atan2(0, 0) = 0 (num_complex::arg's atan2 gives 0 or pi depending on the sign of zero; bins
// that are exactly zero carry no phase information).
#define THZ_ATAN_Q0 -3.333332241e-01f
#define THZ_ATAN_Q1 1.999868155e-01f
#define THZ_ATAN_Q2 -1.425704509e-01f
#define THZ_ATAN_Q3 1.086575910e-01f
#define THZ_ATAN_Q4 -8.009681851e-02f
#define THZ_ATAN_Q5 4.891432077e-02f
#define THZ_ATAN_Q6 -2.002674714e-02f
#define THZ_ATAN_Q7 3.866738873e-03f

// a = min(|x|,|y|) / max(|x|,|y|) in [0, 1]; 0 for x = y = 0 (0 * rcp(tiny) = 0)
__device__ __forceinline__ float atan_octant_ratio(float x, float y)
{
    const float ax = fabsf(x), ay = fabsf(y);
    return fminf(ax, ay) * fast_rcp(fmaxf(fmaxf(ax, ay), 1e-37f));
}
// from atan(a) of the octant back to the angle of (x, y)
__device__ __forceinline__ float atan_unfold(float r, float y, float x)
{
    r = (fabsf(y) > fabsf(x)) ? 1.57079632679489662f - r : r;
    r = (x < 0.0f) ? 3.14159265358979324f - r : r;
    return copysignf(r, y);
}

__device__ __forceinline__ float fast_atan2f(float y, float x)
{
    const float a = atan_octant_ratio(x, y);
    const float z = a * a;
    float q = fmaf(THZ_ATAN_Q7, z, THZ_ATAN_Q6);
    q = fmaf(q, z, THZ_ATAN_Q5);
    q = fmaf(q, z, THZ_ATAN_Q4);
    q = fmaf(q, z, THZ_ATAN_Q3);
    q = fmaf(q, z, THZ_ATAN_Q2);
    q = fmaf(q, z, THZ_ATAN_Q1);
    q = fmaf(q, z, THZ_ATAN_Q0);
    return atan_unfold(fmaf(a * z, q, a), y, x);
}

// four angles at once: same values as four fast_atan2f calls.  The polynomial runs as two packed
// Horner chains written side by side: a dependent v_pk_fma_f32 needs a wait state after its
// producer, and the compiler, left with two calls of a two-angle routine, emitted one chain after
// the other with an s_nop behind every step.
typedef float thz_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void fast_atan2f_x4(const float (&y)[4], const float (&x)[4], float (&r)[4])
{
    const thz_f2 a0 = {atan_octant_ratio(x[0], y[0]), atan_octant_ratio(x[1], y[1])};
    const thz_f2 a1 = {atan_octant_ratio(x[2], y[2]), atan_octant_ratio(x[3], y[3])};
    const thz_f2 z0 = a0 * a0, z1 = a1 * a1;
    thz_f2 q0 = __builtin_elementwise_fma(thz_f2{THZ_ATAN_Q7, THZ_ATAN_Q7}, z0, thz_f2{THZ_ATAN_Q6, THZ_ATAN_Q6});
    thz_f2 q1 = __builtin_elementwise_fma(thz_f2{THZ_ATAN_Q7, THZ_ATAN_Q7}, z1, thz_f2{THZ_ATAN_Q6, THZ_ATAN_Q6});
#define THZ_ATAN_STEP(C)                                            \
    q0 = __builtin_elementwise_fma(q0, z0, thz_f2{C, C});           \
    q1 = __builtin_elementwise_fma(q1, z1, thz_f2{C, C});
    THZ_ATAN_STEP(THZ_ATAN_Q5)
    THZ_ATAN_STEP(THZ_ATAN_Q4)
    THZ_ATAN_STEP(THZ_ATAN_Q3)
    THZ_ATAN_STEP(THZ_ATAN_Q2)
    THZ_ATAN_STEP(THZ_ATAN_Q1)
    THZ_ATAN_STEP(THZ_ATAN_Q0)
#undef THZ_ATAN_STEP
    const thz_f2 r0 = __builtin_elementwise_fma(a0 * z0, q0, a0), r1 = __builtin_elementwise_fma(a1 * z1, q1, a1);
    r[0] = atan_unfold(r0.x, y[0], x[0]);
    r[1] = atan_unfold(r0.y, y[1], x[1]);
    r[2] = atan_unfold(r1.x, y[2], x[2]);
    r[3] = atan_unfold(r1.y, y[3], x[3]);
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
