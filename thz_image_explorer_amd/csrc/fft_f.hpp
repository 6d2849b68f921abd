// fft_f.hpp — "F" family: register-resident three-pass FFT, one trace per
// wavefront, for the trace lengths the benchmarks are quoted on
// (nt = 1024, 2048, 4096  ->  complex N = 512, 1024, 2048).
//
// Decomposition (decimation in frequency, N = R1*R2*R3):
//     n = (N/R1) j1 + m,          m = R3 j2 + j3
//     k = k1 + R1 k2 + R1 R2 k3
//   pass 1  radix-R1 over j1, in registers straight from the coalesced global
//           loads (lane l holds m = C1*l + b), twiddle W_N^(m k1)
//   LDS exchange E1 (XOR-swizzled, conflict free both ways)
//   pass 2  radix-R2 over j2, twiddle W_(R2 R3)^(j3 k2)
//   LDS exchange E2 (XOR-swizzled)
//   pass 3  radix-R3 over j3 -> natural order in LDS
// Each wave owns N cx of LDS (16 KiB at nt = 4096) and never meets a
// workgroup barrier inside its trace loop; the block shares the twiddle
// tables, staged once.  The inverse transform runs through the same passes
// on re<->im swapped data.
#pragma once

#include "thz_device.hpp"

namespace thz {

// Complex values of the F family are 2-element vectors (x = re, y = im): hipcc then
// selects v_pk_add/mul/fma_f32 and folds the re<->im swizzles and sign flips of the
// butterflies into the instructions' op_sel / neg modifiers instead of emitting
// v_mov shuffles (a struct {re, im} costs ~40 % more VALU for a 16-point DFT).
// Layout-compatible with c32.
typedef float cx __attribute__((ext_vector_type(2)));
static_assert(sizeof(cx) == 8, "cx must alias c32");
struct alignas(16) cx2 {
    cx a, b;
};
__device__ __forceinline__ cx cx_mul_pk(cx a, cx b);
__device__ __forceinline__ cx cx_mul(cx a, cx b)
{
#ifdef THZ_CX_MUL_ASM  // A/B builds (scripts/gpu_ab_builds.py): every complex product through cx_mul_pk
    return cx_mul_pk(a, b);
#else
    const cx bs = {-b.y, b.x};
    return a.xx * b + a.yy * bs;
#endif
}
// The same product in two instructions: hipcc builds {-b.y, b.x} with a v_xor and a v_mov in front of every cx_mul
// whose b is not loop-invariant (it folds the broadcasts a.xx / a.yy into op_sel, but not a one-sided negation or a
// swap), i.e. four VALU instructions and a wait state per product.  Here the negation and the swap ride on the
// first instruction's modifiers:  t = {-a.y b.y, a.y b.x};  r = {a.x b.x + t.x, a.x b.y + t.y}.
// A packed-fp32 result needs one wait state before a dependent VALU reads it (the compiler's own listings put an
// s_nop 0 there); inline asm is opaque to its hazard recognizer, so the wait states are written out.
// Same roundings as cx_mul as hipcc compiles it: a.y's products rounded, a.x's fused onto them.
__device__ __forceinline__ cx cx_mul_pk(cx a, cx b)
{
#ifdef THZ_EMU
    const cx bs = {-b.y, b.x};
    return a.xx * b + a.yy * bs;
#else
    cx t, r;
    asm("s_nop 0\n\t"
        "v_pk_mul_f32 %0, %2, %3 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]\n\t"
        "s_nop 0\n\t"
        "v_pk_fma_f32 %1, %2, %3, %0 op_sel_hi:[0,1,1]\n\t"
        "s_nop 0"
        : "=&v"(t), "=v"(r)
        : "v"(a), "v"(b));
    return r;
#endif
}
__device__ __forceinline__ cx cx_conj(cx a) { return cx{a.x, -a.y}; }
__device__ __forceinline__ cx cx_mnegi(cx a) { return cx{a.y, -a.x}; }  // a * (-i)
__device__ __forceinline__ cx cx_swap(cx a) { return cx{a.y, a.x}; }

// ---------------------------------------------------------------- butterflies
// forward (exp(-i...)) DFTs of 4 / 8 / 16 points, natural order in and out

__device__ __forceinline__ void bfly4(cx &a0, cx &a1, cx &a2, cx &a3)
{
    const cx s02 = a0 + a2, d02 = a0 - a2;
    const cx s13 = a1 + a3, d13 = a1 - a3;
    const cx t = cx_mnegi(d13);
    a0 = s02 + s13;
    a1 = d02 + t;
    a2 = s02 - s13;
    a3 = d02 - t;
}

// multiply by exp(-2*pi*i*e/16), e compile-time
template <int E>
__device__ __forceinline__ cx mul_w16(cx v)
{
    constexpr float C1 = 0.92387953251128673848f;  // cos(pi/8)
    constexpr float S1 = 0.38268343236508978178f;  // sin(pi/8)
    constexpr float H = 0.70710678118654752440f;   // cos(pi/4)
    constexpr int e = ((E % 16) + 16) % 16;
    if constexpr (e == 0) return v;
    else if constexpr (e == 4) return cx_mnegi(v);
    else if constexpr (e == 8) return -v;
    else if constexpr (e == 12) return cx{-v.y, v.x};
    else {
        constexpr float cs[16] = {1.0f, C1, H, S1, 0.0f, -S1, -H, -C1, -1.0f, -C1, -H, -S1, 0.0f, S1, H, C1};
        constexpr float sn[16] = {0.0f, S1, H, C1, 1.0f, C1, H, S1, 0.0f, -S1, -H, -C1, -1.0f, -C1, -H, -S1};
        return cx_mul(v, cx{cs[e], -sn[e]});  // exp(-i theta) = (cos, -sin)
    }
}

__device__ __forceinline__ void dft4(cx (&v)[4]) { bfly4(v[0], v[1], v[2], v[3]); }

__device__ __forceinline__ void dft8(cx (&v)[8])
{
    cx e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    cx o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    bfly4(e0, e1, e2, e3);
    bfly4(o0, o1, o2, o3);
    o1 = mul_w16<2>(o1);
    o2 = mul_w16<4>(o2);
    o3 = mul_w16<6>(o3);
    v[0] = (e0 + o0); v[4] = (e0 - o0);
    v[1] = (e1 + o1); v[5] = (e1 - o1);
    v[2] = (e2 + o2); v[6] = (e2 - o2);
    v[3] = (e3 + o3); v[7] = (e3 - o3);
}

__device__ __forceinline__ void dft16(cx (&v)[16])
{
    // A_r[q] = sum_j x[r + 4j] W4^(jq)
    bfly4(v[0], v[4], v[8], v[12]);
    bfly4(v[1], v[5], v[9], v[13]);
    bfly4(v[2], v[6], v[10], v[14]);
    bfly4(v[3], v[7], v[11], v[15]);
    // now v[r + 4q] = A_r[q]; twiddle W16^(r q)
    v[5] = mul_w16<1>(v[5]);   v[6] = mul_w16<2>(v[6]);    v[7] = mul_w16<3>(v[7]);
    v[9] = mul_w16<2>(v[9]);   v[10] = mul_w16<4>(v[10]);  v[11] = mul_w16<6>(v[11]);
    v[13] = mul_w16<3>(v[13]); v[14] = mul_w16<6>(v[14]);  v[15] = mul_w16<9>(v[15]);
    // y[q + 4p] = sum_r W4^(rp) (W16^(rq) A_r[q]) : radix-4 over r inside each q group
    bfly4(v[0], v[1], v[2], v[3]);
    bfly4(v[4], v[5], v[6], v[7]);
    bfly4(v[8], v[9], v[10], v[11]);
    bfly4(v[12], v[13], v[14], v[15]);
    // v[4q + p] = y[q + 4p]  -> transpose to natural order
    cx t;
    t = v[1]; v[1] = v[4]; v[4] = t;
    t = v[2]; v[2] = v[8]; v[8] = t;
    t = v[3]; v[3] = v[12]; v[12] = t;
    t = v[6]; v[6] = v[9]; v[9] = t;
    t = v[7]; v[7] = v[13]; v[13] = t;
    t = v[11]; v[11] = v[14]; v[14] = t;
}

template <int R>
__device__ __forceinline__ void dftR(cx (&v)[R])
{
    static_assert(R == 4 || R == 8 || R == 16, "radix");
    if constexpr (R == 4) dft4(v);
    else if constexpr (R == 8) dft8(v);
    else dft16(v);
}

// Natural-order layout of the spectrum / result in the wave's LDS slice: element
// k sits at nat(k) = k ^ ((k >> 5) & 3).  The XOR only permutes elements inside
// aligned groups of four, and makes every access pattern of the epilogues
// (stride-4 and stride-2 element reads, ascending or mirrored) hit 32 distinct
// 8-byte slots per 32 lanes; index N (the copy of Z[0] / the Nyquist bin) is a
// fixed point.
__device__ __forceinline__ int nat(int k) { return k ^ ((k >> 5) & 3); }

// 16-byte LDS accesses (ds_read_b128 / ds_write_b128): index must be even
__device__ __forceinline__ cx2 ld2(const cx *p) { return *reinterpret_cast<const cx2 *>(p); }
__device__ __forceinline__ void st2(cx *p, cx a, cx b) { *reinterpret_cast<cx2 *>(p) = cx2{a, b}; }

// Compile-time configuration bits of k_f.  Runtime null checks inside the fully
// unrolled passes turn into dozens of tiny basic blocks and make the register
// allocator spill, so everything optional is a template flag instead.
enum : int {
    kCfgAmpPhase = 1,  // also write |X| and the unwrapped phase
    kCfgCMask = 2,     // the per-bin multiplier is complex (K13, reference-pulse Wiener filter)
    kCfgSums = 4,      // the block also sums the stored amplitudes and unwrapped phases of its traces (FSums)
    kCfgBar = 8,       // the block's waves meet at a barrier before each store phase (FArgs::bar says which)
    kCfgBand = 16      // with kCfgCMask: the staged multiplier table covers only the bins FArgs::band_lo4 .. + band_n — where
                       // the real band pass is not zero — between two quads of zeros that every other bin's index is clamped
                       // to.  Half the table at the default 0.2-5 THz: what lets the nt = 4096 chain with the complex
                       // multiplier AND the in-launch sums keep eight waves per block (round 3).
};

// ------------------------------------------------------------------- the plan
template <int R1_, int R2_, int R3_>
struct FPlan {
    static constexpr int R1 = R1_, R2 = R2_, R3 = R3_;
    static constexpr int N = R1 * R2 * R3;       // complex length
    static constexpr int NT = 2 * N;             // real trace length
    static constexpr int M1 = R2 * R3;           // N / R1
    static constexpr int C1 = M1 / kWave;        // m values per lane in pass 1 (1 or 2)
    static constexpr int C2 = R1 * R3 / kWave;   // radix-R2 butterflies per lane
    static constexpr int C3 = R1 * R2 / kWave;   // radix-R3 butterflies per lane
    static constexpr int NG = N / 256;           // 256-bin groups in the store layout
    static_assert(R3 == 8, "pass-2 lane map assumes R3 = 8");
    static_assert(C1 == 1 || C1 == 2, "C1");
    static_assert(C2 >= 1 && C3 >= 1, "lanes must all own a butterfly");
    // LDS per block, in cx: [T1: R1*M1][T2: R2*R3][mask: nf floats, or nf cx][extra][per wave: N + 2]
    static constexpr int T1_ENTRIES = R1 * M1;
    static constexpr int T2_ENTRIES = R2 * R3;
    // N + 1 floats (real multiplier) or N + 1 cx (complex multiplier), padded to 16 bytes; the two
    // floats of padding behind a real mask / the last cx behind a complex one hold the window block bits
    // kCfgBand: [zero quad][up to BAND_BINS bins][zero quad] + the cx that holds the window block bits
    static constexpr int BAND_BINS = N / 2;
    static constexpr int mask_entries(int cfg)
    {
        return (cfg & kCfgBand) ? BAND_BINS + 8 + 2 : (cfg & kCfgCMask) ? N + 2 : (N + 4) / 2;
    }
    static constexpr int WAVE_ENTRIES = N + 2;  // natural order + Z[N] := Z[0], kept 16-byte aligned
    // Small trace-invariant tables the trace loop reads, staged once per block so that no
    // vector-memory load sits between the loop's stores (a load's result can only be waited
    // for together with every store issued before it):
    //   W2N_HEAD  split twiddles w2n[0 .. 255] (the per-lane factors of both epilogues)
    //   WG        w2n[M1 * j], j < R1 (their wave-uniform partners; 256 g = M1 * (256 g / M1))
    //   WIN_SLOTS edge blocks of the time multipliers that are not all ones (f_edge_only)
    static constexpr int W2N_HEAD = 256;
    static constexpr int WG_ENTRIES = 16;
    static constexpr int WIN_BLK = NT / R1;  // floats per window block = 2 C1 * 64
    static constexpr int WIN_SLOTS = (N >= 2048) ? 4 : 6;  // 160 KB LDS leaves room for 4 at nt = 4096
    static constexpr int EXTRA_ENTRIES = W2N_HEAD + WG_ENTRIES + WIN_SLOTS * WIN_BLK / 2;
    static_assert(R1 <= WG_ENTRIES && W2N_HEAD <= N && 256 % M1 == 0, "staged twiddle tables");
    // k_f at nt = 4096 keeps the pass-1 twiddles in COMPACT form (round 3): only the rows k1 = 1, 2, 4, 8 of the
    // lane's even column, T1c[r][lane] = W_N^(C1 lane 2^r) — 2 KiB instead of 16 — and builds the other rows as
    // products (f_core_pass1<P, true>); row 0 of that table is also w2n[4 lane], so the staged head of the split
    // twiddles goes too.  The 16 KiB this frees are the block accumulators of the pixel sums (FSums): the kernel
    // with the sums keeps its eighth wave, the complex-multiplier builds get theirs back.
    static constexpr bool T1_COMPACT = (N == 2048);
    static constexpr int T1C_ROWS = (R1 == 16) ? 4 : 3;
    static constexpr int KF_T1_ENTRIES = T1_COMPACT ? T1C_ROWS * kWave : T1_ENTRIES;
    static constexpr int KF_W2N_HEAD = T1_COMPACT ? 0 : W2N_HEAD;
    static constexpr int KF_EXTRA_ENTRIES = KF_W2N_HEAD + WG_ENTRIES + WIN_SLOTS * WIN_BLK / 2;
    static_assert(!T1_COMPACT || (C1 == 2 && N == 2048), "compact pass-1 twiddles: row 0 must be w2n[4 lane]");
    // kCfgSums: + the block's accumulators, tickets and scratch of FSums behind the wave buffers
    static constexpr int SUM_ENTRIES = (2 * N + 16 + 32) / 2;  // in cx; = FSums::kAreaFloats / 2
    static constexpr size_t lds_bytes(int waves, int cfg = 0)
    {
        return (size_t)(KF_T1_ENTRIES + T2_ENTRIES + mask_entries(cfg) + KF_EXTRA_ENTRIES + waves * WAVE_ENTRIES
                        + ((cfg & kCfgSums) ? SUM_ENTRIES : 0)) * sizeof(cx);
    }

    // E1[k1][m]: column bits 3..4 XORed with k1's low bits
    __device__ static __forceinline__ int e1(int k1, int m) { return k1 * M1 + (m ^ ((k1 & 3) << 3)); }
    // E2 row r = k2*R1 + k1 holds j3 = 0..7; 16-byte unit t of a row sits at t ^ ((r>>2)&3)
    __device__ static __forceinline__ int e2(int r, int j3)
    {
        return r * 8 + ((((j3 >> 1) ^ ((r >> 2) & 3)) << 1) | (j3 & 1));
    }
};

using FPlan4096 = FPlan<16, 16, 8>;
using FPlan2048 = FPlan<8, 16, 8>;
using FPlan1024 = FPlan<8, 8, 8>;

// Host-built tables for one plan (cx arrays in global memory):
//   t1[k1*M1 + C1*l + b ... ] laid out [k1][m]      : W_N^(m k1)
//   t2[k2*8 + j3]                                   : W_(R2*8)^(j3 k2)
//   tl[]  lane constants, see FTables
struct FTables {
    const cx *t1;
    const cx *t2;
    const cx *w2n;  // exp(-i*pi*k/N), k in [0, N): R2C / C2R split twiddles
};

// ------------------------------------------------------------------ the core
// Lane-dependent LDS base indices (cx units), computed once per kernel.  Every
// LDS access of the core is base[...] + compile-time constant, so that it
// becomes one ds_read/ds_write with an immediate offset and no per-access
// address VGPR (without this the ~160 distinct addresses get hoisted out of the
// trace loop and spill).
template <class P>
struct FAddr {
    int w1[4];   // E1 write:  (C1*lane) ^ (q << 3),                q = k1 & 3
    int r1[4];   // E1 read :  (lane>>3)*M1 + (lane&7) + 8*(s ^ q), s = j2 & 3, q = (lane>>3)&3
    int w2[2];   // E2 write:  k1*8 + swz(j3, variant)              (k1 of c2 = 0)
    int r3[4];   // E2 read :  lane*8 + 2*(u ^ ((lane>>2)&3)),      u = j >> 1
    __device__ __forceinline__ void init(int lane)
    {
#pragma unroll
        for (int q = 0; q < 4; ++q) w1[q] = (P::C1 * lane) ^ (q << 3);
        const int ql = (lane >> 3) & 3;
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) r1[s_] = (lane >> 3) * P::M1 + (lane & 7) + 8 * (s_ ^ ql);
        // E2 swizzle operand of row r = k2*R1 + k1 is ((k2*R1/4) + (k1>>2)) & 3; with
        // k1 = (lane>>3) + 8*c2 the lane part is ((lane>>5) + 2*c2) and the k2 part is
        // 0 (R1 = 16) or 2*(k2&1) (R1 = 8): two variants cover everything.
        const int j3 = lane & 7;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int sw = ((lane >> 5) + 2 * v) & 3;
            w2[v] = (lane >> 3) * 8 + ((((j3 >> 1) ^ sw) << 1) | (j3 & 1));
        }
        const int q3 = (lane >> 2) & 3;
#pragma unroll
        for (int u = 0; u < 4; ++u) r3[u] = lane * 8 + 2 * (u ^ q3);
    }
    // once per trace: keep the bases opaque (see launder_v)
    __device__ __forceinline__ void refresh()
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) { w1[i] = launder_v(w1[i]); r1[i] = launder_v(r1[i]); r3[i] = launder_v(r3[i]); }
        w2[0] = launder_v(w2[0]);
        w2[1] = launder_v(w2[1]);
    }
};

// In:  r[c][j1] = z[(N/R1) j1 + C1*lane + c]   (forward data, or swapped data for
//      the inverse).  Out: natural-order spectrum Z[0..N] in `buf` (Z[N] = Z[0]).
// t1/t2 point to the block's LDS copies of the tables.  Ends with wave_sync().
// exp(-2 pi i e / d) for the small angles of the compact twiddle form (|angle| < 0.1): Taylor series in double
constexpr double f_small_cos(double x)
{
    const double z = x * x;
    return 1.0 - z / 2.0 * (1.0 - z / 12.0 * (1.0 - z / 30.0 * (1.0 - z / 56.0 * (1.0 - z / 90.0))));
}
constexpr double f_small_sin(double x)
{
    const double z = x * x;
    return x * (1.0 - z / 6.0 * (1.0 - z / 20.0 * (1.0 - z / 42.0 * (1.0 - z / 72.0 * (1.0 - z / 110.0)))));
}
constexpr double kFTwoPi = 6.28318530717958647692528676655900577;
constexpr float f_unit_re(int e, int d) { return (float)f_small_cos(kFTwoPi * e / d); }
constexpr float f_unit_im(int e, int d) { return (float)-f_small_sin(kFTwoPi * e / d); }

// COMPACT (k_f at nt = 4096, FPlan::T1_COMPACT): t1 points to T1c[r][lane] = W_N^(2 lane 2^r), r = 0..3, and the
// twiddle of (m, k1) = (2 lane + b, k1) is built as
//     W_N^(2 lane k1) = product of the rows of k1's bits  (at most three multiplies deep: 7 = (1 2) 4, 15 = 7 8)
//     W_N^((2 lane + 1) k1) = that  x  W_N^k1              (a compile-time constant)
// — 26 complex multiplies per pass where the full table costs 15 16-byte LDS reads, and 14 KiB of LDS less.
// PK: twiddle products through cx_mul_pk (two VALU instructions each instead of four)
template <class P, bool COMPACT = false, bool PK = false>
__device__ __forceinline__ void f_core_pass1(cx (&r)[P::C1][P::R1], cx *buf, const cx *t1,
                                             const FAddr<P> &ad, int lane)
{
    constexpr int R1 = P::R1, C1 = P::C1, M1 = P::M1;
    // ---- pass 1
#pragma unroll
    for (int c = 0; c < C1; ++c) {
        dftR<R1>(r[c]);
        THZ_SCHED_FENCE();
    }
    if constexpr (COMPACT) {
        static_assert(C1 == 2 && R1 == 16, "compact pass-1 twiddles are written for the 16 x 16 x 8 plan");
        const cx *tc = t1 + launder_v(lane);
        cx tw[R1];
        tw[1] = tc[0];
        tw[2] = tc[kWave];
        tw[4] = tc[2 * kWave];
        tw[8] = tc[3 * kWave];
        st2(buf + ad.w1[0], r[0][0], r[1][0]);
#pragma unroll
        for (int k1 = 1; k1 < R1; ++k1) {
            if (k1 == 3) tw[3] = cx_mul(tw[1], tw[2]);
            if (k1 == 5) tw[5] = cx_mul(tw[1], tw[4]);
            if (k1 == 6) tw[6] = cx_mul(tw[2], tw[4]);
            if (k1 == 7) tw[7] = cx_mul(tw[3], tw[4]);
            if (k1 > 8) tw[k1] = cx_mul(tw[k1 - 8], tw[8]);
            const cx odd = cx{f_unit_re(k1, P::N), f_unit_im(k1, P::N)};  // W_N^k1
            const cx v0 = cx_mul(r[0][k1], tw[k1]);
            const cx v1 = cx_mul(r[1][k1], cx_mul(odd, tw[k1]));
            st2(buf + ad.w1[k1 & 3] + k1 * M1, v0, v1);  // = e1(k1, 2*lane + {0,1})
            if ((k1 & 3) == 3) THZ_SCHED_FENCE();
        }
        wave_sync();
        return;
    }
    const cx *t1l = t1 + launder_v(C1 * lane);
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) {
        if constexpr (C1 == 2) {
            cx v0 = r[0][k1], v1 = r[1][k1];
            if (k1 > 0) {
                const cx2 w = ld2(t1l + k1 * M1);
                v0 = PK ? cx_mul_pk(v0, w.a) : cx_mul(v0, w.a);
                v1 = PK ? cx_mul_pk(v1, w.b) : cx_mul(v1, w.b);
            }
            st2(buf + ad.w1[k1 & 3] + k1 * M1, v0, v1);  // = e1(k1, 2*lane + {0,1})
        } else {
            cx v = r[0][k1];
            if (k1 > 0) v = PK ? cx_mul_pk(v, t1l[k1 * M1]) : cx_mul(v, t1l[k1 * M1]);
            buf[ad.w1[k1 & 3] + k1 * M1] = v;  // = e1(k1, lane)
        }
        if ((k1 & 3) == 3) THZ_SCHED_FENCE();
    }
    wave_sync();
}

// Pass 1 with the lane's twiddles W_N^(lane k1), k1 = 1 .. R1 - 1, already in registers (C1 = 1: they do not depend on
// the transform; a kernel that runs many transforms per table load keeps them) — products through cx_mul_pk
template <class P>
__device__ __forceinline__ void f_core_pass1_regs(cx (&r)[1][P::R1], cx *buf, const cx (&tw)[P::R1], const FAddr<P> &ad)
{
    static_assert(P::C1 == 1, "one column per lane");
    constexpr int R1 = P::R1, M1 = P::M1;
    dftR<R1>(r[0]);
    THZ_SCHED_FENCE();
    buf[ad.w1[0]] = r[0][0];
#pragma unroll
    for (int k1 = 1; k1 < R1; ++k1) {
        buf[ad.w1[k1 & 3] + k1 * M1] = cx_mul_pk(r[0][k1], tw[k1]);  // = e1(k1, lane)
        if ((k1 & 3) == 3) THZ_SCHED_FENCE();
    }
    wave_sync();
}

// KEEP (C3 = 1): the transform's outputs Z[lane + 64 k3] stay in `keep` instead of going to LDS in natural order — for
// a consumer that only reduces over them
template <class P, bool PK = false, bool KEEP = false>
__device__ __forceinline__ void f_core_pass23(cx *buf, const cx *t2, const FAddr<P> &ad, int lane, cx (*keep)[P::R3] = nullptr)
{
    constexpr int R1 = P::R1, R2 = P::R2, R3 = P::R3, C2 = P::C2, C3 = P::C3;
    constexpr int M1 = P::M1;
    // ---- pass 2: lane owns (k1, j3) = ((lane>>3) + 8*c2, lane&7)
    cx b[C2][R2];
#pragma unroll
    for (int c2 = 0; c2 < C2; ++c2) {
#pragma unroll
        for (int j2 = 0; j2 < R2; ++j2)
            b[c2][j2] = buf[ad.r1[j2 & 3] + (8 * (j2 & ~3) + 8 * c2 * M1)];  // = e1(k1, 8*j2 + j3)
        THZ_SCHED_FENCE();
    }
    wave_sync();
    const cx *t2l = t2 + launder_v(lane & 7);
#pragma unroll
    for (int c2 = 0; c2 < C2; ++c2) {
        dftR<R2>(b[c2]);
        THZ_SCHED_FENCE();
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) {
            cx v = b[c2][k2];
            if (k2 > 0) v = PK ? cx_mul_pk(v, t2l[k2 * 8]) : cx_mul(v, t2l[k2 * 8]);
            // row = k2*R1 + k1, k1 = (lane>>3) + 8*c2
            constexpr int kq = R1 / 4;
            const int variant = (((k2 * kq) & 3) >> 1) ^ c2;  // ((k2*R1/4) + 2*c2) & 3 is 0 or 2
            buf[ad.w2[variant & 1] + (k2 * R1 * 8 + 8 * c2 * 8)] = v;  // = e2(row, j3)
            if ((k2 & 3) == 3) THZ_SCHED_FENCE();
        }
    }
    wave_sync();
    // ---- pass 3: lane owns row r3 = lane + 64*c3  (k1 = r3 % R1, k2 = r3 / R1)
    cx d[C3][R3];
#pragma unroll
    for (int c3 = 0; c3 < C3; ++c3) {
#pragma unroll
        for (int u = 0; u < R3 / 2; ++u) {
            const cx2 v = ld2(buf + ad.r3[u] + 512 * c3);  // = e2(row, 2u), e2(row, 2u + 1)
            d[c3][2 * u] = v.a;
            d[c3][2 * u + 1] = v.b;
        }
        THZ_SCHED_FENCE();
    }
    wave_sync();
    // k = lane + off, off = 64*c3 + R1*R2*k3 (a multiple of 64):
    // (k >> 5) & 3 = ((lane >> 5) + (off >> 5)) & 3, and (off >> 5) & 3 is 0 or 2
    const int nb0 = launder_v(lane ^ ((lane >> 5) & 3));
    const int nb1 = launder_v(lane ^ (((lane >> 5) + 2) & 3));
    if constexpr (KEEP) {
        static_assert(C3 == 1 && R1 * R2 == kWave, "outputs lane + 64 k3");
        dftR<R3>(d[0]);
#pragma unroll
        for (int k3 = 0; k3 < R3; ++k3) (*keep)[k3] = d[0][k3];
        return;  // every lane's reads of this pass are behind the wave_sync above: the buffer is free
    }
#pragma unroll
    for (int c3 = 0; c3 < C3; ++c3) {
        dftR<R3>(d[c3]);
#pragma unroll
        for (int k3 = 0; k3 < R3; ++k3) {
            const int off = kWave * c3 + R1 * R2 * k3;
            buf[(((off >> 5) & 2) ? nb1 : nb0) + off] = d[c3][k3];  // = nat(lane + off)
        }
        THZ_SCHED_FENCE();
    }
    if (lane == 0) buf[P::N] = d[0][0];  // Z[N] := Z[0], so that Z[N-k] needs no wrap at k = 0
    wave_sync();
}

// --------------------------------------------------------- R2C / C2R algebra
// Both directions take the split twiddle as w2 = -(i/2) w, w = exp(-i*pi*k/N): the kernels fold the
// factor into the per-lane table when they stage it in LDS (f_stage_w2n).  With
//   s = Z[k] + conj(Z[N-k]),  d = Z[k] - conj(Z[N-k])
// the forward split is  X[k] = s/2 + d w2,  conj(X[N-k]) = s/2 - d w2  — two packed adds, one complex
// multiply and four FMAs; written with E = s/2, O = -(i/2) d as before, the rotation by -i cost
// a move and a sign flip per bin pair on top (the compiler does not fold a swap-and-negate into the
// consumer's op_sel/neg modifiers).  The products are the same real products; only which of the two
// is rounded before the fused add changes.
__device__ __forceinline__ cx f_stage_w2n(cx w) { return cx{0.5f * w.y, -0.5f * w.x}; }

// Compact form (FPlan::T1_COMPACT): no staged head of the split twiddles; w2n[4 l] = W_N^(2 l) is row 0 of the
// compact pass-1 table, and the staged value of w2n[4 l + e], e = 0..3, is that times the constant
// -(i/2) exp(-i pi e / N).
template <int N, int E>
__device__ __forceinline__ cx f_w2n_from_row0(cx t)
{
    if constexpr (E == 0) return f_stage_w2n(t);
    else {
        // -(i/2) (x + i y) = y/2 - i x/2 with (x, y) = exp(-2 pi i E / (2 N))
        constexpr float kx = 0.5f * f_unit_im(E, 2 * N), ky = -0.5f * f_unit_re(E, 2 * N);
        return cx_mul(cx{kx, ky}, t);
    }
}

// Written on p = a + b, m = a - b (s = {p.x, m.y}, d = {m.x, p.y}): the conjugate never exists as a
// value, the multiply takes its splats straight from p and m, and X[N-k] comes out unconjugated —
// a conj() of a packed value is a negate plus a move here, as is building {b.x, -b.y}.
__device__ __forceinline__ void r2c_pair(cx a, cx b, cx w2, cx &xk, cx &xn)
{
    const cx p = a + b, m = a - b;
    const cx bs = {-w2.y, w2.x};
    const cx t = m.xx * w2 + p.yy * bs;  // d * w2
    xk = cx{fmaf(p.x, 0.5f, t.x), fmaf(m.y, 0.5f, t.y)};
    xn = cx{fmaf(p.x, 0.5f, -t.x), fmaf(-m.y, 0.5f, t.y)};  // conj(s/2 - t)
}

// swap(Z'[k]) — the inverse core takes its input with re/im exchanged — from X[k], X[N-k] (as stored,
// unconjugated) and wc = conj(w2):  Z' = E + i (D conj(w)) = E + 2 D wc  with E = {p.x, m.y},
// D = {m.x, p.y}, p = X[k] + X[N-k], m = X[k] - X[N-k].  Same idea as r2c_pair: no conjugate is ever
// built, and the two FMAs write the exchanged pair directly.
__device__ __forceinline__ cx c2r_swapped(cx xk, cx xn, cx wc)
{
    const cx p = xk + xn, m = xk - xn;
    const cx bs = {-wc.y, wc.x};
    const cx t = m.xx * wc + p.yy * bs;  // D * wc
    return cx{fmaf(2.0f, t.y, m.y), fmaf(2.0f, t.x, p.x)};
}

}  // namespace thz

// ----------------------------------------------------------------------------
// F kernels
// ----------------------------------------------------------------------------
namespace thz {

#ifdef THZ_EMU
__device__ __forceinline__ void store_f4(float *p, float a, float b, float c, float d)
{
    p[0] = a; p[1] = b; p[2] = c; p[3] = d;
}
__device__ __forceinline__ void load_f4(const float *p, float &a, float &b, float &c, float &d)
{
    a = p[0]; b = p[1]; c = p[2]; d = p[3];
}
#else
// 16-byte vector access that only promises 4-byte alignment: spectrum rows are
// nf = nt/2+1 elements long, so their starts are not 16-byte aligned.  gfx950
// runs with unaligned access mode on; this still issues one dwordx4.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void store_f4(float *p, float a, float b, float c, float d)
{
    f4u v = {a, b, c, d};
    *reinterpret_cast<f4u *>(p) = v;
}
__device__ __forceinline__ void load_f4(const float *p, float &a, float &b, float &c, float &d)
{
    const f4u v = *reinterpret_cast<const f4u *>(p);
    a = v.x; b = v.y; c = v.z; d = v.w;
}
#endif

struct FArgs {
    size_t npix;
    const float *in;        // (npix, nt) raw traces            [fwd, pipeline]
    const float *pre_win;   // (nt); may be null only when pre_blocks == 0
    cx *fft_out;           // (npix, nf), required
    float *amp_out;         // (npix, nf), required when the kernel is built with AMP_PHASE
    float *ph_out;          // (npix, nf), required when the kernel is built with AMP_PHASE
    const float *mask;      // (nf), required (a vector of ones when no band-pass is wanted)
    const cx *cmask;        // (nf) complex per-bin multiplier, required when the kernel is built with kCfgCMask;
                            // the staged table is cmask[k] * mask[k]
    int bar;                // block barriers in the trace loop: bit 0 before the spectrum stores, bit 1 before
                            // the time stores, bit 2: with a workgroup fence (__syncthreads) instead of a bare s_barrier
    const cx *fft_in;      // (npix, nf)                        [inv only]
    const float *post_win;  // (nt); may be null only when post_blocks == 0
    float *data_out;        // (npix, nt) final trace            [inv, pipeline]
    float *img;             // (npix) or null
    int band_lo4, band_n;   // kCfgBand: first bin (a multiple of 4) and number of bins (a multiple of 4, <= P::BAND_BINS) of
                            // the staged complex multiplier; the real mask is zero at every bin outside
    float *sum_partial;     // (gridDim.x, 2 nf): every block's sums of its traces' stored amplitudes | unwrapped
                            // phases, written whole by the block (zeros if it had no trace); kCfgSums only
};

enum : int { kFwd = 0, kInv = 1, kPipe = 2 };
constexpr int kFBarDefault = 3;  // FArgs::bar of every launch unless THZ_F_BAR says otherwise (measured: +8-11 % on the forward and inverse kernels, 0-5 % on the fused chain)

// Loads one trace's samples for this lane: raw[j1][4] (C1 = 2) or raw[j1][2].
template <class P>
__device__ __forceinline__ void f_load_raw(const float *__restrict__ x, int lane,
                                           float (&raw)[P::R1][2 * P::C1])
{
#pragma unroll
    for (int j1 = 0; j1 < P::R1; ++j1) {
        if constexpr (P::C1 == 2) {
            const float4 v = *reinterpret_cast<const float4 *>(x + 4 * (kWave * j1 + lane));
            raw[j1][0] = v.x; raw[j1][1] = v.y; raw[j1][2] = v.z; raw[j1][3] = v.w;
        } else {
            const float2 v = *reinterpret_cast<const float2 *>(x + 2 * (kWave * j1 + lane));
            raw[j1][0] = v.x; raw[j1][1] = v.y;
        }
    }
}

// Same register layout for a spectrum row (inverse kernel): raw[j][..] = X[C1*(64 j + lane) + c]
// as {re, im} pairs.  Rows are only 8-byte aligned (nf is odd).
template <class P>
__device__ __forceinline__ void f_load_spec(const cx *__restrict__ X, int lane,
                                            float (&raw)[P::R1][2 * P::C1])
{
#pragma unroll
    for (int j = 0; j < P::R1; ++j) {
        const float *src = reinterpret_cast<const float *>(X + P::C1 * (kWave * j + lane));
        if constexpr (P::C1 == 2) {
            load_f4(src, raw[j][0], raw[j][1], raw[j][2], raw[j][3]);
        } else {
            const float2 v = *reinterpret_cast<const float2 *>(src);
            raw[j][0] = v.x; raw[j][1] = v.y;
        }
    }
}

template <class P>
__device__ __forceinline__ void f_load_win(const float *__restrict__ w, int lane, int j1,
                                           float (&out)[2 * P::C1])
{
    if constexpr (P::C1 == 2) {
        const float4 v = *reinterpret_cast<const float4 *>(w + 4 * (kWave * j1 + lane));
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    } else {
        const float2 v = *reinterpret_cast<const float2 *>(w + 2 * (kWave * j1 + lane));
        out[0] = v.x; out[1] = v.y;
    }
}

// Edge block j of a time multiplier: from its LDS slot when it was staged, else from memory.
template <class P, bool STAGED_ONLY>
__device__ __forceinline__ void f_win_block(const float *__restrict__ w, const float *win_s, int slot, int lane,
                                            int j1, float (&out)[2 * P::C1])
{
    if (STAGED_ONLY || slot >= 0) {
        const float *src = win_s + slot * P::WIN_BLK + 2 * P::C1 * lane;
        if constexpr (P::C1 == 2) {
            const float4 v = *reinterpret_cast<const float4 *>(src);
            out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
        } else {
            const float2 v = *reinterpret_cast<const float2 *>(src);
            out[0] = v.x; out[1] = v.y;
        }
    } else {
        f_load_win<P>(w, lane, j1, out);
    }
}

// slot of edge block j (0, R1-2 or R1-1) in a packed slot word: 4 bits each, 15 = not staged
template <class P>
__device__ __forceinline__ int f_slot_of(uint32_t slots, int j)
{
    const int e = j == 0 ? 0 : (j == P::R1 - 2 ? 1 : 2);
    const int v = (int)((slots >> (4 * e)) & 15u);
    return v == 15 ? -1 : v;
}

// kCfgBand: entry of bin k in the staged table [zero quad][bins lo4 .. lo4 + n)[zero quad] — every bin outside the band
// lands on a zero (v_add + v_med3)
__device__ __forceinline__ int f_band_index(int k_minus_lo4_plus4, int n_plus4)
{
    const int hi = k_minus_lo4_plus4 < n_plus4 ? k_minus_lo4_plus4 : n_plus4;
    return hi > 0 ? hi : 0;
}

// Spectrum epilogue.  buf holds Z[0..N] in the nat() layout; on return it holds
// the (unmasked) spectrum X[0..N] in the same layout.
//   groups g <  NG/2: the lane owns bins k = 256 g + 4 lane + c AND their mirrors
//                     N - k: one R2C split per pair gives X[k] and X[N-k], both
//                     written back in place (each Z element is read exactly once);
//                     the low bins are finished here (|X|, arg, mask, stores, scan)
//   lane 0          : X[N/2] = conj(Z[N/2])
//   groups g >= NG/2: bins k = 256 g + 4 lane + c are read back as X and finished
//   bin N           : lane 0, after the last group
// so the unwrap scan always runs over ascending bins.
//
// CMASK: `mask` points to nf complex multipliers H[k]; the stored spectrum is X H (imaginary part of
// bin 0 and of the Nyquist bin forced to 0: the C2R precondition, math_tools.rs:510-512), the stored
// amplitude |X H| (taken before the forcing); phases are those of X, as with the real band pass
// (band_pass_fd.rs:184-212 leaves them alone).  buf keeps the unmultiplied X.
// kCfgSums — pixel sums of the ifft stage's amplitudes and phases (math_tools.rs:427-440: numerators of the
// pixel means) inside the fused launch.  The block keeps ONE set of accumulators in LDS (2 N floats: amplitude and
// phase sums of the bins 0 .. N-1) and its waves add to it in turn: after every 256-bin group of the epilogue a wave
// waits for its ticket of that group — round * W + wave, kept in an LDS counter per group — reads the group's
// accumulators (two 16-byte reads per lane), adds its four amplitudes and four unwrapped phases, writes them back
// and hands the ticket on.  No atomics (LDS float atomics doubled the kernel's time), no barriers (a barrier per
// group put the block's waves of a memory-bound kernel in lock step: +2.5 ms per Mi traces): the waves only have to
// come by in order, which staggers them by one short read-modify-write and otherwise leaves them a whole round of
// slack (wave 0 of round r + 1 waits for the last wave of round r).  The order of every bin's additions is fixed —
// waves 0 .. W-1 of round 0, then round 1, ...; the blocks' rows in order in the final reduction — so the sums are
// deterministic.  The flag accesses and fences are LDS-scoped (lds_flag_load / lds_flag_store): nobody waits for
// its global stores.  Waves without a trace in the block's last round are the last ones of the order and just stay
// away.  The bin N (Nyquist) of every trace is summed per wave in lane 0 and combined once, after the trace loop.
template <class P>
struct FSums {
    static constexpr int N = P::N, NG = P::NG;
    static constexpr int kTickets = 16;                                  // >= NG, keeps the area 16-byte sized
    static constexpr int kAreaFloats = 2 * N + kTickets + 2 * 16;       // sums | tickets | bin-N scratch (16 waves)
    static_assert(NG <= kTickets, "one ticket counter per group");
    float nyq_a, nyq_p;    // lane 0: bin N
    float *area;           // [amplitude sums N][phase sums N][tickets][scratch]
    unsigned round;        // the block's trace round this wave is in
    int wib, wpb;
    unsigned give_up;      // spin bound hit (never, unless a wave of the block died): poisons bin 0 of the row
    // zeroes the area; the caller's __syncthreads() follows
    static __device__ __forceinline__ void clear(float *area, int tid, int nthreads)
    {
        for (int i = tid; i < kAreaFloats; i += nthreads) area[i] = 0.0f;
    }
    __device__ __forceinline__ void init(float *area_, int wave_in_block, int waves_per_block)
    {
        nyq_a = nyq_p = 0.0f;
        area = area_;
        round = 0u;
        wib = wave_in_block;
        wpb = waves_per_block;
        give_up = 0u;
    }
#ifndef THZ_EMU
    typedef float f4e __attribute__((ext_vector_type(4)));
    f4e early_a, early_p;
    unsigned early_t;
#endif
    // issues the visit's LDS reads (ticket + the group's accumulators); group() consumes them
    __device__ __forceinline__ void begin(int g, int lane)
    {
#ifndef THZ_EMU
        const unsigned *tick = reinterpret_cast<const unsigned *>(area + 2 * N) + g;
        const float *sa = area + 256 * g + 4 * lane;
        asm volatile("" ::: "memory");
        early_t = __hip_atomic_load(tick, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
        early_a = *reinterpret_cast<const f4e *>(sa);
        early_p = *reinterpret_cast<const f4e *>(sa + N);
        asm volatile("" ::: "memory");
#else
        (void)g; (void)lane;
#endif
    }
    // adds this wave's group-g values of its current trace to the block's sums, in ticket order
    __device__ __forceinline__ void group(int g, const float (&a)[4], const float (&y)[4], int lane)
    {
        unsigned *tick = reinterpret_cast<unsigned *>(area + 2 * N) + g;
        const unsigned mine = round * (unsigned)wpb + (unsigned)wib;
        unsigned spins = 0u;
        float *sa = area + 256 * g + 4 * lane, *sp = sa + N;
#ifdef THZ_EMU
        while (lds_flag_load(tick) != mine) {
            spin_pause();
            if (++spins > (1u << 24)) {  // ~ seconds: something else is badly wrong; do not hang the GPU over it
                give_up = 1u;
                break;
            }
        }
        const float4 va = *reinterpret_cast<const float4 *>(sa), vp = *reinterpret_cast<const float4 *>(sp);
        *reinterpret_cast<float4 *>(sa) = make_float4(va.x + a[0], va.y + a[1], va.z + a[2], va.w + a[3]);
        *reinterpret_cast<float4 *>(sp) = make_float4(vp.x + y[0], vp.y + y[1], vp.z + y[2], vp.w + y[3]);
        wave_sync();  // every lane's update is issued before lane 0 hands the ticket on
        if (lane == 0) lds_flag_store(tick, mine + 1u);
#else
        // One LDS round trip per visit instead of three (round 3).  The CU's LDS executes a wave's DS instructions
        // in issue order (what wave_sync() relies on everywhere in this file), so (a) the accumulators can be read
        // right behind the ticket, in the same wait: if the ticket read shows this wave's number, the previous
        // owner's updates — issued before its ticket write — were executed before it, and the reads behind it see
        // them; otherwise the values are dropped and read again; (b) the hand-over needs no wait for the updates to
        // land: the ticket write is issued behind them and executes behind them.  Only the compiler has to keep the
        // order, hence the barriers.
        typedef float f4v __attribute__((ext_vector_type(4)));
        // the first attempt's reads were issued by begin(), in front of the group's arctangents and unwrap: their
        // latency is behind the wave when it gets here (a stale ticket only costs the re-read below)
        f4v va = early_a, vp = early_p;
        unsigned t = early_t;
        for (;;) {
            if (THZ_UNIFORM((int)t) == (int)mine) break;
            t = __hip_atomic_load(tick, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
            va = *reinterpret_cast<const f4v *>(sa);  // (plain loads: a volatile access through the generic pointer
            vp = *reinterpret_cast<const f4v *>(sp);  //  loses the LDS address space and becomes a flat load)
            asm volatile("" ::: "memory");
            if (THZ_UNIFORM((int)t) == (int)mine) break;
            spin_pause();
            if (++spins > (1u << 24)) {  // ~ seconds: something else is badly wrong; do not hang the GPU over it
                give_up = 1u;
                break;
            }
        }
        *reinterpret_cast<float4 *>(sa) = make_float4(va.x + a[0], va.y + a[1], va.z + a[2], va.w + a[3]);
        *reinterpret_cast<float4 *>(sp) = make_float4(vp.x + y[0], vp.y + y[1], vp.z + y[2], vp.w + y[3]);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) __hip_atomic_store(tick, mine + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
#endif
    }
    // after the trace loop (and a block barrier): bin N across the waves, then the block's row of sum_partial
    __device__ __forceinline__ void finish(float *row, int nf, int lane)
    {
        float *scratch = area + 2 * N + kTickets;
        if (lane == 0) {
            scratch[wib] = nyq_a;
            scratch[16 + wib] = nyq_p;
            if (give_up) area[0] = __builtin_nanf("");
        }
        block_lds_barrier();  // all waves are through their last group and have left their bin-N sums
        if (wib == 0 && lane == 0) {
            float a = 0.0f, ph = 0.0f;
            for (int w = 0; w < wpb; ++w) {
                a += scratch[w];
                ph += scratch[16 + w];
            }
            row[nf - 1] = a;
            row[2 * nf - 1] = ph;
        }
        for (int i = wib * kWave + lane; i < N; i += wpb * kWave) {
            row[i] = area[i];
            row[nf + i] = area[N + i];
        }
    }
};

// WC: w2n_s is row 0 of the compact pass-1 table instead of the staged head of the split twiddles
// STORE_FFT = false (fused chain): the masked spectrum is stored by f_inverse_input, which forms the same products
// X m (X H) anyway and holds them two adjacent bins per lane — sixteen 1 KiB stores in one short burst instead of two
// per group spread over the whole epilogue; only the Nyquist bin is still stored here.
template <class P, bool AMP_PHASE, bool CMASK = false, bool SUMS = false, bool WC = false, bool STORE_FFT = true, bool BAND = false>
__device__ __forceinline__ void f_spectrum_epilogue(cx *buf, const cx *w2n_s, const cx *wg_s,
                                                    const float *mask, size_t p, const FArgs &A,
                                                    int lane, FSums<P> *sums = nullptr)
{
    static_assert(!BAND || CMASK, "the band-limited table is the complex multiplier's");
    const int band_off = 4 - A.band_lo4, band_cap = A.band_n + 4;  // BAND: table entry of bin k = clamp(k + band_off, 0, band_cap)
    static_assert(!SUMS || AMP_PHASE, "the sums are those of the amplitudes and phases");
    constexpr int N = P::N, NG = P::NG;
    static_assert(NG % 2 == 0, "pair ownership splits the groups in halves");
    const int nf = N + 1;
    const float kPi = 3.14159274101257324219f, kTwoPi = 2.0f * kPi;
    constexpr bool want_phase = AMP_PHASE;
    cx wl[4];
    if constexpr (WC) {
        const cx t = w2n_s[lane];
        wl[0] = f_w2n_from_row0<N, 0>(t);
        wl[1] = f_w2n_from_row0<N, 1>(t);
        wl[2] = f_w2n_from_row0<N, 2>(t);
        wl[3] = f_w2n_from_row0<N, 3>(t);
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) wl[c] = w2n_s[4 * lane + c];
    }
    float carry = 0.0f;       // sum of adjusted differences of all previous groups
    float prev_tail = 0.0f;   // raw phase of the last bin of the previous group
    float first = 0.0f;       // raw phase of bin 0
    float last_unwrapped = 0.0f, last_raw = 0.0f;
    // nat(256 g + 4 lane + c) = 256 g + fb[c];  nat(N - 256 g - 4 lane - c) = mb[c] - 256 g
    int fb[4], mb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        fb[c] = launder_v(nat(4 * lane + c));
        mb[c] = launder_v(nat(N - 4 * lane - c));
    }
    const int kb = launder_v(4 * lane);
#pragma unroll 1
    for (int g = 0; g < NG; ++g) {
        const int k0 = 256 * g + kb;
        cx *zf = buf + 256 * g;
        cx X[4];
        if (g < NG / 2) {
            cx *zm = buf - 256 * g;
            const cx wg = wg_s[(256 / P::M1) * g];  // w2n[256 g], wave-uniform
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const cx a = zf[fb[c]], b = zm[mb[c]];
                cx xk, xn;
                // g = 0: wg = w2n[0] = (1, 0) and the product is wl[c] itself — no select needed
                r2c_pair(a, b, cx_mul(wl[c], wg), xk, xn);
                X[c] = xk;
                zf[fb[c]] = xk;  // X[k]
                zm[mb[c]] = xn;  // X[N-k]  (k = 0: buf[N] = X[N], and X[0] over Z[0])
            }
            if (g == NG / 2 - 1) {
                // the one bin without a partner: X[N/2] = conj(Z[N/2]); every lane has
                // finished its reads of this group before lane 0 rewrites the slot
                wave_sync();
                if (lane == 0) buf[nat(N / 2)] = cx_conj(buf[nat(N / 2)]);
                wave_sync();
            }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) X[c] = zf[fb[c]];
        }
        float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // the stored amplitudes (AMP_PHASE)
        if constexpr (CMASK) {
            cx Y[4];
            {
                // BAND: the quad's entries, or — outside the band — one of the two quads of zeros (k0, band_lo4 and band_n
                // are multiples of 4, so a quad is inside or outside as a whole)
                const cx *hm = reinterpret_cast<const cx *>(mask) + (BAND ? f_band_index(k0 + band_off, band_cap) : k0);  // LDS copy
                const cx2 h01 = ld2(hm), h23 = ld2(hm + 2);
                // cx_mul_pk: the multiplier comes out of LDS every time, so hipcc's cx_mul spends two extra instructions per
                // product on {-h.y, h.x} (A/B, alternating builds in one process: 14.80 -> 14.53 ms with the multiplier,
                // 15.99 -> 15.75 with multiplier and sums; the kernels without a complex multiplier gain nothing from it)
                Y[0] = cx_mul_pk(X[0], h01.a); Y[1] = cx_mul_pk(X[1], h01.b);
                Y[2] = cx_mul_pk(X[2], h23.a); Y[3] = cx_mul_pk(X[3], h23.b);
            }
            if constexpr (AMP_PHASE) {
#pragma unroll
                for (int c = 0; c < 4; ++c) a[c] = fast_sqrt(fmaf(Y[c].x, Y[c].x, Y[c].y * Y[c].y));
                store_f4(A.amp_out + p * nf + k0, a[0], a[1], a[2], a[3]);
            }
            if constexpr (STORE_FFT) {
                if (g == 0 && lane == 0) Y[0].y = 0.0f;  // bin 0
                float *f = reinterpret_cast<float *>(A.fft_out + p * nf + k0);
                store_f4(f, Y[0].x, Y[0].y, Y[1].x, Y[1].y);
                store_f4(f + 4, Y[2].x, Y[2].y, Y[3].x, Y[3].y);
            }
        } else {
            float m[4];
            {
                const float4 mv = *reinterpret_cast<const float4 *>(mask + k0);  // LDS copy
                m[0] = mv.x; m[1] = mv.y; m[2] = mv.z; m[3] = mv.w;
            }
            if constexpr (AMP_PHASE) {
#pragma unroll
                for (int c = 0; c < 4; ++c) a[c] = fast_sqrt(fmaf(X[c].x, X[c].x, X[c].y * X[c].y)) * m[c];
                store_f4(A.amp_out + p * nf + k0, a[0], a[1], a[2], a[3]);
            }
            if constexpr (STORE_FFT) {
                float *f = reinterpret_cast<float *>(A.fft_out + p * nf + k0);
                store_f4(f, X[0].x * m[0], X[0].y * m[0], X[1].x * m[1], X[1].y * m[1]);
                store_f4(f + 4, X[2].x * m[2], X[2].y * m[2], X[3].x * m[3], X[3].y * m[3]);
            }
        }
        if constexpr (want_phase) {
            float ph[4];
            // the sums' LDS reads (ticket + accumulators) go out in front of the arctangents: by the time group() looks at
            // them, ~150 instructions later, their latency is behind the wave (measured: 14.28 -> 14.15 ms per Mi traces)
            if constexpr (SUMS) sums->begin(g, lane);
            {
                const float yy[4] = {X[0].y, X[1].y, X[2].y, X[3].y}, xx[4] = {X[0].x, X[1].x, X[2].x, X[3].x};
                fast_atan2f_x4(yy, xx, ph);
            }
            {
                // unconditional read + select: a branch here splits the block between the two packed
                // chains of the arctangent and they end up one after the other again
                const float lane0 = wave_bcast<0>(ph[0]);
                first = g == 0 ? lane0 : first;
            }
            float prev = wave_shr1(ph[3]);
            if (lane == 0) prev = prev_tail;
            float s_[4];
            float run = 0.0f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float d = ph[c] - (c == 0 ? prev : ph[c - 1]);
                d += (d > kPi) ? -kTwoPi : ((d < -kPi) ? kTwoPi : 0.0f);  // numpy_unwrap's if / else if
                if (g == 0 && c == 0 && lane == 0) d = 0.0f;
                run += d;
                s_[c] = run;
            }
            const float incl = wave_scan_add(run);
            const float excl = wave_shr1(incl);
            const float base = carry + excl;
            float y[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) y[c] = first + (base + s_[c]);
            store_f4(A.ph_out + p * nf + k0, y[0], y[1], y[2], y[3]);
            carry += wave_bcast<kWave - 1>(incl);
            prev_tail = wave_bcast<kWave - 1>(ph[3]);
            if (g == NG - 1) {
                last_unwrapped = wave_bcast<kWave - 1>(y[3]);
                last_raw = prev_tail;
            }
            if constexpr (SUMS) sums->group(g, a, y, lane);
        }
    }
    // Nyquist bin k = N (real): lane 0
    if (lane == 0) {
        const float xr = buf[N].x;
        float aN;
        if constexpr (CMASK) {
            const cx hN = reinterpret_cast<const cx *>(mask)[BAND ? f_band_index(N + band_off, band_cap) : N];
            const cx yN = cx{xr * hN.x, xr * hN.y};
            A.fft_out[p * nf + N] = cx{yN.x, 0.0f};
            aN = fast_sqrt(fmaf(yN.x, yN.x, yN.y * yN.y));
        } else {
            const float mN = mask[N];
            A.fft_out[p * nf + N] = cx{xr * mN, 0.0f};
            aN = fabsf(xr) * mN;
        }
        if constexpr (AMP_PHASE) A.amp_out[p * nf + N] = aN;
        if constexpr (want_phase) {
            const float phn = fast_atan2f(0.0f, xr);
            float d = phn - last_raw;
            d += (d > kPi) ? -kTwoPi : ((d < -kPi) ? kTwoPi : 0.0f);
            A.ph_out[p * nf + N] = last_unwrapped + d;
            if constexpr (SUMS) {
                sums->nyq_a += aN;
                sums->nyq_p += last_unwrapped + d;
            }
        }
    }
}

// Builds the (swapped) input of the inverse core in the core layout,
// r[c][j1] <- swap(Z'[n]), n = M1 j1 + C1 lane + c, from the spectrum X[0..N]
// held in buf in the nat() layout (X[N] real at buf[N]).  MASKED: multiply by
// the band-pass mask on the way (fused chain; the stored copy is unmasked so
// that the phases of the whole spectrum could be taken).
// CMASK (with MASKED): the multiplier is complex, mask points to nf cx; X[0] H[0] and X[N] H[N] lose their
// imaginary parts after the multiply.
// STORE (fused chain, with MASKED): fft_row = the trace's row of the spectrum output; the masked bins n < N — the
// values the inverse transform is built from, to the bit — are stored from here (bin N: the spectrum epilogue).
template <class P, bool MASKED, bool CMASK = false, bool WC = false, bool STORE = false, bool BAND = false>
__device__ __forceinline__ void f_inverse_input(const cx *buf, const cx *w2n_s, const cx *wg_s,
                                                const float *__restrict__ mask, int lane,
                                                cx (&r)[P::C1][P::R1], cx *fft_row = nullptr, int band_off = 0, int band_cap = 0)
{
    // the masked values are stored AND consumed: no product of this function may be fused into the split's adds
    // (the stand-alone inverse must land on the same samples from the stored spectrum, bit for bit)
#pragma clang fp contract(off)
    static_assert(!STORE || MASKED, "the stored spectrum is the masked one");
    constexpr int N = P::N, R1 = P::R1, C1 = P::C1, M1 = P::M1;
    cx wlc[C1];  // conj of the lane's staged twiddles: the product below is conj(w2) directly
    if constexpr (WC) {
        // w2n[2 lane + c] = w2n[4 (lane >> 1)] x exp(-i pi (2 (lane & 1) + c) / N): row 0 of the compact table at
        // lane >> 1, times one of two constants
        static_assert(C1 == 2, "compact twiddles: 16 x 16 x 8 plan");
        const cx t = w2n_s[lane >> 1];
        const bool odd = (lane & 1) != 0;
        const cx e0 = f_w2n_from_row0<N, 0>(t), e1 = f_w2n_from_row0<N, 1>(t);
        const cx e2 = f_w2n_from_row0<N, 2>(t), e3 = f_w2n_from_row0<N, 3>(t);
        wlc[0] = cx_conj(odd ? e2 : e0);
        wlc[1] = cx_conj(odd ? e3 : e1);
    } else {
#pragma unroll
        for (int c = 0; c < C1; ++c) wlc[c] = cx_conj(w2n_s[C1 * lane + c]);
    }
    // n = M1 j1 + C1 lane + c and its mirror N - n; M1 is a multiple of 32*4 only for
    // C1 = 2 (M1 = 128), so (n >> 5) & 3 is lane-constant there; for C1 = 1 (M1 = 64)
    // it alternates with j1 & 1 -> two base variants cover both plans.
    int fbase[2][C1], mbase[2][C1];
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int c = 0; c < C1; ++c) {
            fbase[v][c] = launder_v(nat(M1 * v + C1 * lane + c) - M1 * v);
            mbase[v][c] = launder_v(nat(N - M1 * v - C1 * lane - c) + M1 * v);
        }
    constexpr int TOP = M1 * (R1 - 1) + C1 - 1;
    const int mk_f = launder_v(C1 * lane);                 // mask[n]     = mask[mk_f + M1 j1 + c]
    const int mk_r = launder_v(N - TOP - C1 * lane);       // mask[N - n] = mask[mk_r + TOP - (M1 j1 + c)]
    float *frow = reinterpret_cast<float *>(fft_row) + 2 * mk_f;
#pragma unroll
    for (int j1 = 0; j1 < R1; ++j1) {
        const cx wgc = cx_conj(wg_s[j1]);  // conj(w2n[M1 j1]), wave-uniform
        cx kept[C1];
#pragma unroll
        for (int c = 0; c < C1; ++c) {
            const int off = M1 * j1 + c;
            const cx wc = j1 == 0 ? wlc[c] : cx_mul(wlc[c], wgc);
            cx xk = buf[fbase[j1 & 1][c] + M1 * j1];
            cx xn = buf[mbase[j1 & 1][c] - M1 * j1];
            if (off == 0 && !(MASKED && CMASK)) {
                // n == 0 only in lane 0: X[0] and X[N] are real (realfft ignores /
                // rejects their imaginary parts, SURVEY a'-4)
                if (lane == 0) {
                    xk.y = 0.0f;
                    xn.y = 0.0f;
                }
            }
            if constexpr (MASKED && CMASK) {
                const cx *hm = reinterpret_cast<const cx *>(mask);
                if constexpr (BAND) {
                    xk = cx_mul_pk(xk, hm[f_band_index(mk_f + band_off + off, band_cap)]);
                    xn = cx_mul_pk(xn, hm[f_band_index(mk_r + band_off + (TOP - off), band_cap)]);
                } else {
                    xk = cx_mul_pk(xk, hm[mk_f + off]);
                    xn = cx_mul_pk(xn, hm[mk_r + (TOP - off)]);
                }
                if (off == 0 && lane == 0) {
                    xk.y = 0.0f;
                    xn.y = 0.0f;
                }
            } else if constexpr (MASKED) {
                const float mk = mask[mk_f + off], mn = mask[mk_r + (TOP - off)];
                xk = cx{xk.x * mk, xk.y * mk};
                xn = cx{xn.x * mn, xn.y * mn};
            }
            kept[c] = xk;
            r[c][j1] = c2r_swapped(xk, xn, wc);
        }
        if constexpr (STORE) {
            if constexpr (C1 == 2) store_f4(frow + 2 * M1 * j1, kept[0].x, kept[0].y, kept[1].x, kept[1].y);
            else *reinterpret_cast<float2 *>(frow + 2 * M1 * j1) = make_float2(kept[0].x, kept[0].y);
        }
        if ((j1 & 1) == 1) THZ_SCHED_FENCE();
    }
}

// Window block bits: bit j set <=> samples [j*NT/R1, (j+1)*NT/R1) of a time
// multiplier hold a value != 1.  Each block scans the (L2-resident) vector once
// in its prologue.  The usual multipliers are edge tapers (the reference's
// adapted-Blackman windows and Time Band Pass), i.e. only block 0 and the last
// two blocks are set: those launches read 3 small pieces of the table per trace
// instead of all of it.  Anything else takes the "full" path.
template <class P>
__device__ __forceinline__ bool f_edge_only(uint32_t blocks)
{
    constexpr uint32_t edge = 1u | (1u << (P::R1 - 1)) | (1u << (P::R1 - 2));
    return (blocks & ~edge) == 0;
}

// does block j need its multiply?  (wave-uniform; FULL is a compile-time copy)
template <class P, bool FULL>
__device__ __forceinline__ bool f_block_on(uint32_t blocks, int j)
{
    if constexpr (FULL) return true;
    else if (j == 0 || j >= P::R1 - 2) return ((blocks >> j) & 1u) != 0;
    else return false;
}

template <class P, bool WIN_FULL, bool STAGED_ONLY = false>
__device__ __forceinline__ void f_time_epilogue(const cx *buf, size_t p, const FArgs &A,
                                                uint32_t post_blocks, const float *win_s, uint32_t post_slots,
                                                int lane)
{
    constexpr int NT = P::NT, R1 = P::R1, C1 = P::C1;
    const float fnt = (float)NT;
    const float *post_w = launder_uniform(A.post_win);
    // n = C1*(64 j + lane) + c: for C1 = 2 the pair (2u, 2u+1), u = 64 j + lane, is one
    // 16-byte unit also under nat(): unit (2u ^ (s & 2)), elements swapped when s & 1,
    // s = (lane >> 4) & 3.  For C1 = 1 s = ((lane >> 5) + 2 j) & 3: two variants.
    const int s2 = (lane >> 4) & 3;
    const int ob2 = launder_v((2 * lane) ^ (s2 & 2));
    const bool swap2 = (s2 & 1) != 0;
    const int ob1a = launder_v(nat(lane)), ob1b = launder_v(nat(kWave + lane) - kWave);
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < R1; ++j) {
        float v[2 * C1];
        if constexpr (C1 == 2) {
            const cx2 rr = ld2(buf + ob2 + 2 * kWave * j);
            const cx e0 = swap2 ? rr.b : rr.a, e1 = swap2 ? rr.a : rr.b;
            v[0] = e0.y / fnt; v[1] = e0.x / fnt;
            v[2] = e1.y / fnt; v[3] = e1.x / fnt;
        } else {
            const cx rr = buf[((j & 1) ? ob1b : ob1a) + kWave * j];
            v[0] = rr.y / fnt;
            v[1] = rr.x / fnt;
        }
        if (f_block_on<P, WIN_FULL>(post_blocks, j)) {
            float w[2 * C1];
            if constexpr (WIN_FULL) f_load_win<P>(post_w, lane, j, w);
            else f_win_block<P, STAGED_ONLY>(post_w, win_s, f_slot_of<P>(post_slots, j), lane, j, w);
#pragma unroll
            for (int i = 0; i < 2 * C1; ++i) v[i] *= w[i];
        }
#pragma unroll
        for (int i = 0; i < 2 * C1; ++i) acc += v[i] * v[i];
        float *o = A.data_out + p * NT;
        if constexpr (C1 == 2)
            *reinterpret_cast<float4 *>(o + 4 * (kWave * j + lane)) = make_float4(v[0], v[1], v[2], v[3]);
        else
            *reinterpret_cast<float2 *>(o + 2 * (kWave * j + lane)) = make_float2(v[0], v[1]);
        if ((j & 1) == 1) THZ_SCHED_FENCE();
    }
    if (A.img) {
        acc = wave_reduce_add(acc);
        if (lane == 0) A.img[p] = acc;
    }
}

// Makes the prefetched next trace land *now*.  On gfx9 a vector-memory load can only be waited
// for with vmcnt(0) while stores of the same wave are in flight (the compiler treats loads and
// stores in the one counter as unordered), so the wait is taken right before a store phase
// starts — when the previous stores have long completed — instead of at the top of the next
// trace, where it would drain the stores just issued.
template <class P>
__device__ __forceinline__ void f_land_prefetch(float (&raw)[P::R1][2 * P::C1])
{
#pragma unroll
    for (int j = 0; j < P::R1; ++j)
#pragma unroll
        for (int i = 0; i < 2 * P::C1; ++i) raw[j][i] = launder_f(raw[j][i]);
}

// the waves of a block only align their store phases here; they share no data, so no fence is needed
__device__ __forceinline__ void f_block_barrier(int mode)
{
#ifdef THZ_EMU
    (void)mode;
    __syncthreads();
#else
    if (mode & 4) __syncthreads();
    else __builtin_amdgcn_s_barrier();
#endif
}

struct FTrue { static constexpr bool value = true; };
struct FFalse { static constexpr bool value = false; };

template <class P, int MODE, int CFG>
__global__ __launch_bounds__(512) void k_f(FArgs A, FTables T)
{
    THZ_DYN_LDS(lds);
    constexpr int N = P::N, NT = P::NT, R1 = P::R1, C1 = P::C1;
    constexpr bool AMP_PHASE = (CFG & kCfgAmpPhase) != 0;
    constexpr bool CMASK = (CFG & kCfgCMask) != 0 && MODE != kInv;
    constexpr bool SUMS = (CFG & kCfgSums) != 0;
    constexpr bool BAND = (CFG & kCfgBand) != 0 && CMASK;
    static_assert(!SUMS || (MODE == kPipe && AMP_PHASE && (CFG & kCfgBar) != 0), "sums: fused chain, block-uniform trace loop");
    constexpr int ME = P::mask_entries(CFG);
    const int nf = N + 1;
    const int lane = lane_id();
    const int wib = THZ_UNIFORM((int)(threadIdx.x >> 6));
    const int wpb = (int)(blockDim.x >> 6);
    constexpr bool TC = P::T1_COMPACT;  // compact pass-1 twiddles (FPlan): t1 = T1c[r][lane], no staged w2n head
    cx *t1 = reinterpret_cast<cx *>(lds);
    cx *t2 = t1 + P::KF_T1_ENTRIES;
    float *mask_s = reinterpret_cast<float *>(t2 + P::T2_ENTRIES);
    cx *w2n_s = TC ? t1 : t2 + P::T2_ENTRIES + ME;  // TC: row 0 of T1c is w2n[4 lane]
    cx *wg_s = t2 + P::T2_ENTRIES + ME + P::KF_W2N_HEAD;
    float *win_s = reinterpret_cast<float *>(wg_s + P::WG_ENTRIES);
    cx *buf = t2 + P::T2_ENTRIES + ME + P::KF_EXTRA_ENTRIES + (size_t)wib * P::WAVE_ENTRIES;
    if constexpr (!TC) {
        for (int i = (int)threadIdx.x; i < P::W2N_HEAD; i += (int)blockDim.x) w2n_s[i] = f_stage_w2n(T.w2n[i]);
    }
    if ((int)threadIdx.x < R1) wg_s[threadIdx.x] = T.w2n[P::M1 * (int)threadIdx.x];
    if constexpr (TC) {
        for (int i = (int)threadIdx.x; i < P::KF_T1_ENTRIES; i += (int)blockDim.x)
            t1[i] = T.t1[(1 << (i / kWave)) * P::M1 + C1 * (i % kWave)];  // row 2^r, column C1 lane
    } else {
        for (int i = (int)threadIdx.x; i < P::T1_ENTRIES; i += (int)blockDim.x) t1[i] = T.t1[i];
    }
    for (int i = (int)threadIdx.x; i < P::T2_ENTRIES; i += (int)blockDim.x) t2[i] = T.t2[i];
    if constexpr (BAND) {
        // [zero quad][bins band_lo4 .. band_lo4 + band_n)[zero quad]: the same products as the full table holds there
        cx *cm = reinterpret_cast<cx *>(mask_s);
        for (int i = (int)threadIdx.x; i < A.band_n + 8; i += (int)blockDim.x) {
            const int k = A.band_lo4 + i - 4;
            cx v = cx{0.0f, 0.0f};
            if (i >= 4 && i < A.band_n + 4 && k < nf) {
                const float m = A.mask[k];
                const cx h = A.cmask[k];
                v = cx{h.x * m, h.y * m};
            }
            cm[i] = v;
        }
    } else if constexpr (CMASK) {
        cx *cm = reinterpret_cast<cx *>(mask_s);
        for (int i = (int)threadIdx.x; i < nf; i += (int)blockDim.x) {
            const float m = A.mask[i];
            const cx h = A.cmask[i];
            cm[i] = cx{h.x * m, h.y * m};
        }
    } else if (MODE != kInv) {
        for (int i = (int)threadIdx.x; i < nf; i += (int)blockDim.x) mask_s[i] = A.mask[i];
    }
    float *sum_area = reinterpret_cast<float *>(t2 + P::T2_ENTRIES + ME + P::KF_EXTRA_ENTRIES + (size_t)wpb * P::WAVE_ENTRIES);
    if constexpr (SUMS) {
        static_assert(FSums<P>::kAreaFloats == 2 * P::SUM_ENTRIES, "LDS size of the sums area");
        FSums<P>::clear(sum_area, (int)threadIdx.x, (int)blockDim.x);
    }
    // window block bits (see f_edge_only), kept in the padding behind the mask
    unsigned int *bits = reinterpret_cast<unsigned int *>(mask_s + (2 * ME - 2));
    if (threadIdx.x < 2) bits[threadIdx.x] = 0u;
    __syncthreads();
    {
        constexpr int BLK = NT / R1;
        unsigned int mine_pre = 0u, mine_post = 0u;
        for (int i = (int)threadIdx.x; i < NT; i += (int)blockDim.x) {
            if (MODE != kInv && A.pre_win && A.pre_win[i] != 1.0f) mine_pre |= 1u << (i / BLK);
            if (MODE != kFwd && A.post_win && A.post_win[i] != 1.0f) mine_post |= 1u << (i / BLK);
        }
        if (mine_pre) atomicOr(&bits[0], mine_pre);
        if (mine_post) atomicOr(&bits[1], mine_post);
    }
    __syncthreads();
    const uint32_t pre_blocks = THZ_UNIFORM((int)bits[0]);
    const uint32_t post_blocks = THZ_UNIFORM((int)bits[1]);
    const bool pre_edge = f_edge_only<P>(pre_blocks);
    const bool post_edge = f_edge_only<P>(post_blocks);
    // LDS slots for the edge blocks that are on (4 bits per candidate, 15 = stays in memory)
    uint32_t pre_slots = 0xfffu, post_slots = 0xfffu;
    {
        int next = 0;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const int j = e == 0 ? 0 : (e == 1 ? R1 - 2 : R1 - 1);
            if (MODE != kInv && pre_edge && ((pre_blocks >> j) & 1u) && next < P::WIN_SLOTS) {
                pre_slots = (pre_slots & ~(15u << (4 * e))) | ((uint32_t)next << (4 * e));
                for (int i = (int)threadIdx.x; i < P::WIN_BLK; i += (int)blockDim.x)
                    win_s[next * P::WIN_BLK + i] = A.pre_win[j * P::WIN_BLK + i];
                ++next;
            }
        }
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const int j = e == 0 ? 0 : (e == 1 ? R1 - 2 : R1 - 1);
            if (MODE != kFwd && post_edge && ((post_blocks >> j) & 1u) && next < P::WIN_SLOTS) {
                post_slots = (post_slots & ~(15u << (4 * e))) | ((uint32_t)next << (4 * e));
                for (int i = (int)threadIdx.x; i < P::WIN_BLK; i += (int)blockDim.x)
                    win_s[next * P::WIN_BLK + i] = A.post_win[j * P::WIN_BLK + i];
                ++next;
            }
        }
    }
    __syncthreads();

    FAddr<P> ad;
    ad.init(lane);
    FSums<P> sums;
    if constexpr (SUMS) sums.init(sum_area, wib, wpb);
    const size_t stride = (size_t)gridDim.x * wpb;
    size_t p = (size_t)blockIdx.x * wpb + wib;
    float raw[R1][2 * C1];
    float x_nyq_next = 0.0f;
    if (p < A.npix) {
        if constexpr (MODE != kInv) {
            f_load_raw<P>(A.in + p * NT, lane, raw);
        } else {
            f_load_spec<P>(A.fft_in + p * nf, lane, raw);
            x_nyq_next = A.fft_in[p * nf + N].x;
        }
    }

    // The trace loop exists twice: FAST when every time multiplier is an edge taper whose
    // blocks all sit in LDS (the default chain) — then the loop issues no vector-memory load
    // other than the prefetch of the next trace — and the general form.  They are separate
    // loops rather than branches inside one because the compiler merges the memory-counter
    // state of all paths at every join: with a memory fallback anywhere in the loop, the common
    // path waits for vmcnt(0) at the top of each trace, i.e. for all of its own stores.
    // The first trace's loads are made to land before the loop is entered: at the loop header
    // the compiler merges "entered from above" with "came around the back edge", and with loads
    // pending on entry (nothing issued after them) it would wait for vmcnt(0) at the top of
    // every trace — the back edge has this trace's 16+ stores behind the prefetch, which need
    // not have completed.
    f_land_prefetch<P>(raw);
    // BAR (kCfgBar): the block's waves meet at a barrier before each store phase, so that the
    // eight adjacent rows a block owns in every output array are written within one short window
    // (scripts/probe_shapes.hip, S8 / S9: the same bytes move 8 % faster that way).  The loop then has
    // a block-uniform trip count; a wave without a trace in the last round only takes the barriers.
    auto trace_loop = [&](auto fast_tag, auto bar_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    constexpr bool BAR = decltype(bar_tag)::value;
    const float *mask_l = launder_uniform((const float *)mask_s);
    auto part_a = [&]() {
        cx r[C1][R1];
        ad.refresh();
        if constexpr (MODE != kInv) {
            const float *pre_w = launder_uniform(A.pre_win);
            // window, then hand the samples to pass 1
            if (FAST || pre_edge) {
#pragma unroll
                for (int j1 = 0; j1 < R1; ++j1) {
                    if (f_block_on<P, false>(pre_blocks, j1)) {
                        float w[2 * C1];
                        f_win_block<P, FAST>(pre_w, win_s, f_slot_of<P>(pre_slots, j1), lane, j1, w);
#pragma unroll
                        for (int i = 0; i < 2 * C1; ++i) raw[j1][i] *= w[i];
                    }
                }
            } else {
#pragma unroll
                for (int j1 = 0; j1 < R1; ++j1) {
                    float w[2 * C1];
                    f_load_win<P>(pre_w, lane, j1, w);
#pragma unroll
                    for (int i = 0; i < 2 * C1; ++i) raw[j1][i] *= w[i];
                    if ((j1 & 3) == 3) THZ_SCHED_FENCE();
                }
            }
#pragma unroll
            for (int j1 = 0; j1 < R1; ++j1) {
#pragma unroll
                for (int c = 0; c < C1; ++c) r[c][j1] = cx{raw[j1][2 * c], raw[j1][2 * c + 1]};
            }
            // The next trace of this wave is prefetched into registers right
            // after pass 1 (of the inverse transform in the fused chain): `r` is dead
            // by then, so the 16 KiB of prefetch registers never coexist with it.
            f_core_pass1<P, TC>(r, buf, t1, ad, lane);
            if (MODE == kFwd && p + stride < A.npix) f_load_raw<P>(A.in + (p + stride) * NT, lane, raw);
            f_core_pass23<P>(buf, t2, ad, lane);
            if (MODE == kFwd) f_land_prefetch<P>(raw);
        } else {
            // prefetched spectrum -> LDS in the nat() layout, X[N] at buf[N]
            {
                const int s2 = (lane >> 4) & 3;
                const int sb2 = launder_v((2 * lane) ^ (s2 & 2));
                const bool swap2 = (s2 & 1) != 0;
                const int sb1a = launder_v(nat(lane)), sb1b = launder_v(nat(kWave + lane) - kWave);
#pragma unroll
                for (int j = 0; j < R1; ++j) {
                    if constexpr (C1 == 2) {
                        const cx e0 = cx{raw[j][0], raw[j][1]}, e1 = cx{raw[j][2], raw[j][3]};
                        st2(buf + sb2 + 2 * kWave * j, swap2 ? e1 : e0, swap2 ? e0 : e1);
                    } else {
                        buf[((j & 1) ? sb1b : sb1a) + kWave * j] = cx{raw[j][0], raw[j][1]};
                    }
                }
                if (lane == 0) buf[N] = cx{x_nyq_next, 0.0f};
            }
            wave_sync();
            f_inverse_input<P, false, false, TC>(buf, launder_uniform((const cx *)w2n_s), launder_uniform((const cx *)wg_s),
                                                 nullptr, lane, r);
            wave_sync();
            f_core_pass1<P, TC>(r, buf, t1, ad, lane);
            if (p + stride < A.npix) {
                f_load_spec<P>(A.fft_in + (p + stride) * nf, lane, raw);
                x_nyq_next = A.fft_in[(p + stride) * nf + N].x;
            }
            f_core_pass23<P>(buf, t2, ad, lane);
        }
    };
    // spectrum stores (+ the inverse transform of the fused chain)
    auto part_b = [&]() {
        if constexpr (MODE != kInv) {
            f_spectrum_epilogue<P, AMP_PHASE, CMASK, SUMS, TC, MODE != kPipe, BAND>(buf, launder_uniform((const cx *)w2n_s),
                                                                              launder_uniform((const cx *)wg_s), mask_l, p, A, lane,
                                                                              &sums);
            if constexpr (MODE == kPipe) {
                cx r[C1][R1];
                f_inverse_input<P, true, CMASK, TC, true, BAND>(buf, launder_uniform((const cx *)w2n_s), launder_uniform((const cx *)wg_s),
                                                                mask_l, lane, r, A.fft_out + p * nf, 4 - A.band_lo4, A.band_n + 4);
                wave_sync();  // every lane has read Z before the core overwrites buf
                f_core_pass1<P, TC>(r, buf, t1, ad, lane);
                if (p + stride < A.npix) f_load_raw<P>(A.in + (p + stride) * NT, lane, raw);
                f_core_pass23<P>(buf, t2, ad, lane);
            }
        }
    };
    // time-domain stores
    auto part_c = [&]() {
        if constexpr (MODE != kFwd) {
            f_land_prefetch<P>(raw);
            if constexpr (MODE == kInv) x_nyq_next = launder_f(x_nyq_next);
            if (FAST) f_time_epilogue<P, false, true>(buf, p, A, post_blocks, win_s, post_slots, lane);
            else if (post_edge) f_time_epilogue<P, false>(buf, p, A, post_blocks, win_s, post_slots, lane);
            else f_time_epilogue<P, true>(buf, p, A, post_blocks, win_s, post_slots, lane);
        }
        wave_sync();
    };
    if constexpr (!BAR) {
        for (; p < A.npix; p += stride) {
            part_a();
            part_b();
            part_c();
        }
    } else {
        const size_t p_first = (size_t)blockIdx.x * wpb;  // the block's wave 0 has the most traces
        const size_t iters = p_first < A.npix ? (A.npix - p_first + stride - 1) / stride : 0;
        const int bar = THZ_UNIFORM(A.bar);
        for (size_t it = 0; it < iters; ++it, p += stride) {
            const bool on = p < A.npix;
            if (on) part_a();
            if (MODE != kInv && (bar & 1)) f_block_barrier(bar);
            if constexpr (SUMS) sums.round = (unsigned)it;
            if (on) part_b();
            if (MODE != kFwd && (bar & 2)) f_block_barrier(bar);
            if (on) part_c();
        }
    }
    };
    bool all_staged = true;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const int j = e == 0 ? 0 : (e == 1 ? R1 - 2 : R1 - 1);
        if (MODE != kInv && ((pre_blocks >> j) & 1u) && ((pre_slots >> (4 * e)) & 15u) == 15u) all_staged = false;
        if (MODE != kFwd && ((post_blocks >> j) & 1u) && ((post_slots >> (4 * e)) & 15u) == 15u) all_staged = false;
    }
    if constexpr ((CFG & kCfgBar) != 0) {
        if (pre_edge && post_edge && all_staged) trace_loop(FTrue{}, FTrue{});
        else trace_loop(FFalse{}, FTrue{});
        if constexpr (SUMS) sums.finish(A.sum_partial + (size_t)blockIdx.x * (size_t)(2 * nf), nf, lane);
    } else {
        if (pre_edge && post_edge && all_staged) trace_loop(FTrue{}, FFalse{});
        else trace_loop(FFalse{}, FFalse{});
    }
}

}  // namespace thz
