// fft_p.hpp — "P" family: the fused default chain for trace lengths that are not a power of two
// but factor into three small radices, nt = R1 R2 R3 — above all nt = 1001 = 7 x 11 x 13, the
// length of every real scan the reference opens (io.rs:576-628 hands it straight to realfft,
// math_tools.rs:375, which has no cliff at such lengths).
//
// One wave per PAIR of real traces, packed as one complex transform z = x1 + i x2 of length nt
// (the spectra are separated with Z[nt-k], the inverse transforms conj(Y1 + i Y2): the same
// packing as the chirp-z kernels of fft_fb.hpp) — but the transform itself is a direct
// three-pass mixed-radix FFT, decimation in frequency:
//     n = (N/R1) j1 + m,          m = R3 j2 + j3
//     k = k1 + R1 k2 + R1 R2 k3
//   pass 1  radix-R1 over j1 straight from the global loads, twiddle W_N^(m k1)
//   pass 2  radix-R2 over j2, in place, twiddle W_(R2 R3)^(j3 k2)
//   pass 3  radix-R3 over j3 -> natural order
// with N/R butterflies of radix R dealt to the 64 lanes in rounds (143 / 91 / 77 butterflies for
// 1001 = 7 x 11 x 13: 3 / 2 / 2 rounds).  Small DFTs of odd length use the symmetric direct form
// (x_j +- x_(R-j) against cos / sin constants: (R-1)^2 / 2 packed FMAs), even lengths split once by
// radix 2.  One length-N complex transform costs ~600 VALU instructions per wave against ~3000 for
// the two 2048-point transforms of the chirp-z convolution it replaces (fft_fb.hpp: 0.17 of the
// HBM roofline, VALU-bound).
#pragma once

#include "fft_f.hpp"

#include <utility>

namespace thz {

// ----------------------------------------------------------------- compile-time cos / sin
constexpr double kPiD = 3.14159265358979323846264338327950288;
constexpr double p_cos_series(double x)
{
    double t = 1.0, s = 1.0;
    for (int n = 1; n < 24; ++n) {
        t *= -x * x / (double)((2 * n - 1) * (2 * n));
        s += t;
    }
    return s;
}
constexpr double p_sin_series(double x)
{
    double t = x, s = x;
    for (int n = 1; n < 24; ++n) {
        t *= -x * x / (double)((2 * n) * (2 * n + 1));
        s += t;
    }
    return s;
}
// cos / sin of 2 pi j / R, argument reduced to [-pi, pi]
template <int R>
struct PTrig {
    static constexpr double ang(int j)
    {
        const int jj = ((j % R) + R) % R;
        return 2.0 * kPiD * (double)(jj <= R / 2 ? jj : jj - R) / (double)R;
    }
    static constexpr float c(int j) { return (float)p_cos_series(ang(j)); }
    static constexpr float s(int j) { return (float)p_sin_series(ang(j)); }
};

// ----------------------------------------------------------------- small DFTs, natural order in and out
// forward: X[k] = sum_j x[j] exp(-2 pi i j k / R)
template <int R, class = void>
struct PDft;

template <>
struct PDft<1, void> {
    static __device__ __forceinline__ void run(cx (&)[1]) {}
};

// odd R: pairs a_j = x_j + x_(R-j), b_j = x_j - x_(R-j);  X[k], X[R-k] = C_k -+ i S_k with
// C_k = x_0 + sum_j a_j cos(2 pi j k / R), S_k = sum_j b_j sin(2 pi j k / R)
template <int R>
struct PDft<R, std::enable_if_t<(R % 2 == 1 && R > 1)>> {
    static constexpr int H = (R - 1) / 2;
    template <int K, int... J>
    static __device__ __forceinline__ void row(const cx (&a)[H], const cx (&b)[H], cx x0, cx &lo, cx &hi,
                                               std::integer_sequence<int, J...>)
    {
        cx C = x0, S = cx{0.0f, 0.0f};
        ((C = C + a[J] * cx{PTrig<R>::c((J + 1) * K), PTrig<R>::c((J + 1) * K)},
          S = S + b[J] * cx{PTrig<R>::s((J + 1) * K), PTrig<R>::s((J + 1) * K)}),
         ...);
        lo = cx{C.x + S.y, C.y - S.x};  // C - i S
        hi = cx{C.x - S.y, C.y + S.x};  // C + i S
    }
    template <int... K>
    static __device__ __forceinline__ void rows(cx (&v)[R], const cx (&a)[H], const cx (&b)[H], cx x0,
                                                std::integer_sequence<int, K...>)
    {
        (row<K + 1>(a, b, x0, v[K + 1], v[R - 1 - K], std::make_integer_sequence<int, H>{}), ...);
    }
    static __device__ __forceinline__ void run(cx (&v)[R])
    {
        cx a[H], b[H];
        const cx x0 = v[0];
        cx s = x0;
#pragma unroll
        for (int j = 0; j < H; ++j) {
            a[j] = v[j + 1] + v[R - 1 - j];
            b[j] = v[j + 1] - v[R - 1 - j];
            s = s + a[j];
        }
        rows(v, a, b, x0, std::make_integer_sequence<int, H>{});
        v[0] = s;
    }
};

// even R: one radix-2 decimation-in-time step over two DFTs of length R / 2
template <int R>
struct PDft<R, std::enable_if_t<(R % 2 == 0)>> {
    static constexpr int H = R / 2;
    template <int... K>
    static __device__ __forceinline__ void combine(cx (&v)[R], const cx (&e)[H], const cx (&o)[H],
                                                   std::integer_sequence<int, K...>)
    {
        // t = o[K] W_R^K, W = (cos, -sin)
        ((v[K] = e[K] + cx_mul(o[K], cx{PTrig<R>::c(K), -PTrig<R>::s(K)}),
          v[K + H] = e[K] - cx_mul(o[K], cx{PTrig<R>::c(K), -PTrig<R>::s(K)})),
         ...);
    }
    static __device__ __forceinline__ void run(cx (&v)[R])
    {
        cx e[H], o[H];
#pragma unroll
        for (int i = 0; i < H; ++i) {
            e[i] = v[2 * i];
            o[i] = v[2 * i + 1];
        }
        PDft<H>::run(e);
        PDft<H>::run(o);
        combine(v, e, o, std::make_integer_sequence<int, H>{});
    }
};

// ----------------------------------------------------------------- the plan
template <int R1_, int R2_, int R3_>
struct PPlan {
    static constexpr int R1 = R1_, R2 = R2_, R3 = R3_;
    static constexpr int N = R1 * R2 * R3;  // complex length = real trace length
    static constexpr int NF = N / 2 + 1;
    static constexpr int M1 = R2 * R3;
    static constexpr int B1 = M1, B2 = R1 * R3, B3 = R1 * R2;  // butterflies per pass
    static constexpr int rounds(int b) { return (b + kWave - 1) / kWave; }
    static constexpr int RD1 = rounds(B1), RD2 = rounds(B2), RD3 = rounds(B3);
    static constexpr int pad4(int v) { return (v + 3) & ~3; }
    // LDS per block: [T1: N cx, [k1][m]][T2: R2 R3 cx, [k2][j3]][per wave: N cx (+pad)][mask nf floats, or nf cx][pre N][post N]
    static constexpr int T1_ENTRIES = N, T2_ENTRIES = M1 + (M1 & 1);
    static constexpr int WAVE_ENTRIES = N + (N & 1);
    // in-launch pixel sums (PSums): amplitude | phase accumulators over the epilogue's 256-bin groups, tickets, spare
    static constexpr int SUM_BINS = 256 * ((NF + 255) / 256);
    static constexpr int SUM_FLOATS = 2 * SUM_BINS + 16;
    static constexpr size_t lds_bytes(int waves, int pairs_per_wave = 1, bool cmask = false, bool sums = false)
    {
        return (size_t)(T1_ENTRIES + T2_ENTRIES + waves * pairs_per_wave * WAVE_ENTRIES) * sizeof(cx)
               + (size_t)((cmask ? 2 : 1) * pad4(NF) + 2 * pad4(N) + (sums ? SUM_FLOATS : 0)) * sizeof(float);
    }
};

constexpr int kPPairsDefault = 1;  // pairs of traces per wave unless THZ_P_PAIRS says otherwise
using PPlan1001 = PPlan<7, 11, 13>;
using PPlan1000 = PPlan<10, 10, 10>;
using PPlan1200 = PPlan<10, 10, 12>;  // round lengths beyond the reference's sample data (one pair per wave only)
using PPlan1500 = PPlan<10, 10, 15>;
using PPlan2000 = PPlan<10, 10, 20>;

// threads of the largest block a launch of plan P with Q pairs per wave can have (the launcher fits the waves to LDS;
// the table-only configuration fits the most)
// waves of a block: as many as LDS holds — except that 13 to 15 waves become 12: a thirteenth wave puts four waves on
// one SIMD and so costs every wave of the block a quarter of its registers (170 -> 128), which only a full sixteen
// are worth
template <class P>
constexpr int p_block_waves(int q, bool cmask, bool sums)
{
    int w = 16 / q;
    while (w > 1 && P::lds_bytes(w, q, cmask, sums) > (size_t)160 * 1024) --w;
    return (w >= 13 && w < 16) ? 12 : w;
}
template <class P>
constexpr int p_max_threads(int q) { return p_block_waves<P>(q, false, false) * kWave; }

struct PTables {
    const cx *t1;  // [k1][m]  W_N^(m k1),           N entries
    const cx *t2;  // [k2][j3] W_(R2 R3)^(j3 k2),    R2 R3 entries
};

// lane -> butterfly maps of the three passes (trace-invariant; kept per round as LDS base indices).
// Q pairs of traces share a wave (Q = 1 or 2): their Q N / R butterflies of a pass are dealt to the lanes as ONE
// list — pair q's transform lives in the wave's q-th buffer — so that the last round of a pass is fuller
// (1001 = 7 x 11 x 13: 143 / 91 / 77 butterflies are 3 / 2 / 2 rounds for one pair, 286 / 182 / 154 are 5 / 3 / 3
// for two: a quarter fewer radix-R butterflies issued per transform).
template <class P, int Q>
struct PAddr {
    static constexpr int RD1 = P::rounds(Q * P::B1), RD2 = P::rounds(Q * P::B2), RD3 = P::rounds(Q * P::B3);
    int m1[RD1];   // pass 1: m of the butterfly                         (table / window index)
    int g1[RD1];   // pass 1: global sample offset (2 q N + m), floats from the unit's first trace
    int l1[RD1];   // pass 1: LDS base q WE + m
    int b2[RD2];   // pass 2: q WE + k1 M1 + j3   (butterfly b = k1 R3 + j3)
    int t2[RD2];   // pass 2: j3
    int b3[RD3];   // pass 3: q WE + k1 M1 + R3 k2 (butterfly b = k1 + R1 k2)
    int o3[RD3];   // pass 3: q WE + b
    // A lane without a butterfly in a pass's last round works on the butterfly it already finished in round 0
    // (its own, so nobody else is writing there) and does not store: no lane ever reads what another lane writes
    // in the same round — true on the GPU by lock-step anyway, and what keeps the emulation ThreadSanitizer-clean.
    __device__ __forceinline__ void init(int lane)
    {
        static_assert(P::B1 >= kWave && P::B2 >= kWave && P::B3 >= kWave, "round 0 is full in every pass");
        constexpr int WE = P::WAVE_ENTRIES;
#pragma unroll
        for (int i = 0; i < RD1; ++i) {
            int b = lane + kWave * i;
            b = b < Q * P::B1 ? b : lane;
            const int q = b / P::B1, m = b % P::B1;
            m1[i] = m;
            g1[i] = 2 * q * P::N + m;
            l1[i] = q * WE + m;
        }
#pragma unroll
        for (int i = 0; i < RD2; ++i) {
            int b = lane + kWave * i;
            b = b < Q * P::B2 ? b : lane;
            const int q = b / P::B2;
            b -= q * P::B2;
            b2[i] = q * WE + (b / P::R3) * P::M1 + b % P::R3;
            t2[i] = b % P::R3;
        }
#pragma unroll
        for (int i = 0; i < RD3; ++i) {
            int b = lane + kWave * i;
            b = b < Q * P::B3 ? b : lane;
            const int q = b / P::B3;
            b -= q * P::B3;
            b3[i] = q * WE + (b % P::R1) * P::M1 + P::R3 * (b / P::R1);
            o3[i] = q * WE + b;
        }
    }
    __device__ __forceinline__ void refresh()
    {
#pragma unroll
        for (int i = 0; i < RD1; ++i) { m1[i] = launder_v(m1[i]); g1[i] = launder_v(g1[i]); l1[i] = launder_v(l1[i]); }
#pragma unroll
        for (int i = 0; i < RD2; ++i) { b2[i] = launder_v(b2[i]); t2[i] = launder_v(t2[i]); }
#pragma unroll
        for (int i = 0; i < RD3; ++i) { b3[i] = launder_v(b3[i]); o3[i] = launder_v(o3[i]); }
    }
};

// pass 1 of one round: v[j1] holds z[M1 j1 + m]; writes y[k1][m] W_N^(m k1) to buf[lb + k1 M1], lb = q WE + m
template <class P>
__device__ __forceinline__ void p_pass1_round(cx (&v)[P::R1], cx *buf, const cx *t1, int m, int lb, bool on)
{
    PDft<P::R1>::run(v);
    if (on) {
        buf[lb] = v[0];
#pragma unroll
        for (int k1 = 1; k1 < P::R1; ++k1) buf[lb + k1 * P::M1] = cx_mul(v[k1], t1[k1 * P::M1 + m]);
    }
}

// passes 2 and 3 on the wave's Q buffers; leaves the natural-order transforms in buf[q WE + 0 .. N).
// Ends with wave_sync().
template <class P, int Q>
__device__ __forceinline__ void p_pass23(cx *buf, const cx *t2, const PAddr<P, Q> &ad, int lane)
{
    constexpr int R1 = P::R1, R2 = P::R2, R3 = P::R3;
    constexpr int RD2 = PAddr<P, Q>::RD2, RD3 = PAddr<P, Q>::RD3;
    wave_sync();
    // ---- pass 2: butterfly (k1, j3), elements k1 M1 + R3 j2 + j3, in place
#pragma unroll
    for (int i = 0; i < RD2; ++i) {
        const bool on = lane + kWave * i < Q * P::B2;
        cx v[R2];
#pragma unroll
        for (int j2 = 0; j2 < R2; ++j2) v[j2] = buf[ad.b2[i] + R3 * j2];
        PDft<R2>::run(v);
        if (on) {
            buf[ad.b2[i]] = v[0];
#pragma unroll
            for (int k2 = 1; k2 < R2; ++k2) buf[ad.b2[i] + R3 * k2] = cx_mul(v[k2], t2[k2 * R3 + ad.t2[i]]);
        }
        THZ_SCHED_FENCE();
    }
    wave_sync();
    // ---- pass 3: butterfly (k1, k2), elements k1 M1 + R3 k2 + j3 -> X[k1 + R1 k2 + R1 R2 k3]: every
    // round's inputs are read before any output is written (outputs land in other butterflies' inputs)
    cx d[RD3][R3];
#pragma unroll
    for (int i = 0; i < RD3; ++i) {
#pragma unroll
        for (int j3 = 0; j3 < R3; ++j3) d[i][j3] = buf[ad.b3[i] + j3];
    }
    wave_sync();
#pragma unroll
    for (int i = 0; i < RD3; ++i) {
        const bool on = lane + kWave * i < Q * P::B3;
        PDft<R3>::run(d[i]);
        if (on) {
#pragma unroll
            for (int k3 = 0; k3 < R3; ++k3) buf[ad.o3[i] + R1 * R2 * k3] = d[i][k3];
        }
        THZ_SCHED_FENCE();
    }
    wave_sync();
}

// v, or +0.0 when `zero` — on the bits, so that no floating-point identity can bring a sign back
__device__ __forceinline__ float p_zero_if(float v, bool zero)
{
    const uint32_t bits = __builtin_bit_cast(uint32_t, v) & (zero ? 0u : 0xffffffffu);
    return __builtin_bit_cast(float, bits);
}

// In-launch pixel sums of the stored amplitudes and unwrapped phases (the numerators of the ifft stage's pixel means,
// math_tools.rs:427-440), as in the F kernels (fft_f.hpp, FSums): the block keeps one set of accumulators in LDS and
// its waves add to it group by group in TICKET order — wave w's visit of group g in its r-th trip has ticket
// 2 (r W + w) (+ 1 for the pair's second trace) in the group's counter — so there are no atomics, no barriers, and the order of every bin's additions is
// fixed (deterministic sums).  Waves without a pair in the block's last trip are the last ones of the order and stay
// away.
template <class P>
struct PSums {
    static constexpr int SB = P::SUM_BINS;
    float *area;       // [amplitude sums SB][phase sums SB][16 tickets]
    unsigned round;
    int wib, wpb;
    unsigned give_up;
    static __device__ __forceinline__ void clear(float *area, int tid, int nthreads)
    {
        for (int i = tid; i < P::SUM_FLOATS; i += nthreads) area[i] = 0.0f;
    }
    __device__ __forceinline__ void init(float *area_, int wave_in_block, int waves_per_block)
    {
        area = area_; round = 0u; wib = wave_in_block; wpb = waves_per_block; give_up = 0u;
    }
    // adds one trace's group-g values (t = 0 / 1: the pair's first / second trace; live = false: the pair has no second
    // trace, only the ticket moves on).  A wave's two visits hold consecutive tickets, (round W + wave) 2 + t, so the
    // second never waits for another wave — and only one trace's eight values are alive at a time (the kernel sits at
    // the 128-VGPR cap of a 1 024-thread block).
    __device__ __forceinline__ void group(int g, int t, const FBUnwrap &u, bool live, int lane)
    {
        unsigned *tick = reinterpret_cast<unsigned *>(area + 2 * SB) + g;
        const unsigned mine = (round * (unsigned)wpb + (unsigned)wib) * 2u + (unsigned)t;
        unsigned spins = 0u;
        while (lds_flag_load(tick) != mine) {
            spin_pause();
            if (++spins > (1u << 24)) {  // never, unless a wave of the block died: do not hang the GPU over it
                give_up = 1u;
                break;
            }
        }
        if (live) {  // wave-uniform
            float *sa = area + 256 * g + 4 * lane, *sp = sa + SB;
            const float4 va = *reinterpret_cast<const float4 *>(sa), vp = *reinterpret_cast<const float4 *>(sp);
            *reinterpret_cast<float4 *>(sa) = make_float4(va.x + u.a[0], va.y + u.a[1], va.z + u.a[2], va.w + u.a[3]);
            *reinterpret_cast<float4 *>(sp) = make_float4(vp.x + u.y[0], vp.y + u.y[1], vp.z + u.y[2], vp.w + u.y[3]);
        }
        wave_sync();  // every lane's update is issued before lane 0 hands the ticket on
        if (lane == 0) lds_flag_store(tick, mine + 1u);
    }
    // after the trip loop and a block barrier: the block's row of sum_partial
    __device__ __forceinline__ void finish(float *row, int nf, int tid, int nthreads)
    {
        if (give_up && (tid & (kWave - 1)) == 0) area[0] = __builtin_nanf("");
        block_lds_barrier();
        for (int i = tid; i < nf; i += nthreads) {
            row[i] = area[i];
            row[nf + i] = area[SB + i];
        }
    }
};

// Same argument block as the chirp-z kernels (w / bf unused).  A wave takes Q consecutive pairs (a "unit" of
// 2 Q traces) per trip.
// CM: A.cmask holds nf complex multipliers H[k] applied on top of the real mask (stored spectrum X m H, amplitude
// |X m H|, phase of X; see fb_finish_bins_c) — the reference-pulse deconvolution of a real scan in the same launch.
// SUMS (fused chain, Q = 1): the pixel sums of the stored amplitudes and phases in the same launch (PSums).
template <class P, int MODE, int Q, bool CM = false, bool SUMS = false>
__global__ __launch_bounds__(p_max_threads<P>(Q)) void k_p(FBArgs A, PTables T)
{
    static_assert(!SUMS || (MODE == kPipe && Q == 1), "in-launch sums: the fused chain, one pair per wave");
    THZ_DYN_LDS(lds);
    constexpr int N = P::N, NF = P::NF, R1 = P::R1, M1 = P::M1, WE = P::WAVE_ENTRIES;
    constexpr int RD1 = PAddr<P, Q>::RD1;
    const int lane = lane_id();
    const int wib = THZ_UNIFORM((int)(threadIdx.x >> 6));
    const int wpb = (int)(blockDim.x >> 6);
    cx *t1 = reinterpret_cast<cx *>(lds);
    cx *t2 = t1 + P::T1_ENTRIES;
    cx *buf = t2 + P::T2_ENTRIES + (size_t)wib * (Q * WE);
    float *mask_s = reinterpret_cast<float *>(t2 + P::T2_ENTRIES + (size_t)wpb * (Q * WE));
    float *pre_s = mask_s + (CM ? 2 : 1) * P::pad4(NF);
    float *post_s = pre_s + P::pad4(N);
    float *sum_area = post_s + P::pad4(N);
    if constexpr (SUMS) PSums<P>::clear(sum_area, (int)threadIdx.x, (int)blockDim.x);
    for (int i = (int)threadIdx.x; i < P::T1_ENTRIES; i += (int)blockDim.x) t1[i] = T.t1[i];
    for (int i = (int)threadIdx.x; i < M1; i += (int)blockDim.x) t2[i] = T.t2[i];
    if constexpr (CM) {
        cx *cm = reinterpret_cast<cx *>(mask_s);
        for (int i = (int)threadIdx.x; i < NF; i += (int)blockDim.x) {
            const float m = A.mask[i];
            const cx h = A.cmask[i];
            cm[i] = cx{h.x * m, h.y * m};
        }
    } else {
        for (int i = (int)threadIdx.x; i < NF; i += (int)blockDim.x) mask_s[i] = A.mask[i];
    }
    for (int i = (int)threadIdx.x; i < N; i += (int)blockDim.x) {
        // the 1/2 of the spectrum split X1 = (Z[k] + conj Z[N-k]) / 2 rides on the window: exact (a power of two), and
        // eight multiplies per bin quad fewer in the epilogue
        pre_s[i] = 0.5f * (A.pre_win ? A.pre_win[i] : 1.0f);
        post_s[i] = A.post_win ? A.post_win[i] : 1.0f;
    }
    __syncthreads();

    PAddr<P, Q> ad;
    ad.init(lane);
    PSums<P> sums;
    if constexpr (SUMS) sums.init(sum_area, wib, wpb);
    const DivConst by_nt((float)N);
    constexpr int n_groups = (NF + 255) / 256;  // epilogue groups of 256 bins: bin = 256 g + 4 lane + c
    const size_t n_units = (A.npix + 2 * Q - 1) / (2 * Q);
    const size_t stride = (size_t)gridDim.x * wpb;

    for (size_t u = (size_t)blockIdx.x * wpb + wib; u < n_units; u += stride) {
        const size_t p0 = 2 * Q * u;  // first trace of the unit
        if constexpr (SUMS) sums.round = (unsigned)((u - ((size_t)blockIdx.x * wpb + wib)) / stride);
        ad.refresh();
        const cx *t1l = launder_uniform((const cx *)t1);
        const cx *t2l = launder_uniform((const cx *)t2);
        const float *pre_l = launder_uniform((const float *)pre_s);
        const float *post_l = launder_uniform((const float *)post_s);
        const float *mask_l = launder_uniform((const float *)mask_s);
        const int lb4 = launder_v(4 * lane), lb1 = launder_v(lane);
        // traces of the unit that exist (wave-uniform): n_tr in 1 .. 2 Q
        const int n_tr = (int)((A.npix - p0) < (size_t)(2 * Q) ? (A.npix - p0) : (size_t)(2 * Q));

        if constexpr (MODE != kInv) {
            // ---- pass 1 from memory: z[n] = (x1[n] + i x2[n]) pre[n], n = M1 j1 + m.  Every round's loads are
            // issued before the first butterfly (one trip to HBM per unit); a trace that does not exist is read
            // from the unit's first trace instead (in bounds) and enters as zero
            const float *x0 = A.in + p0 * (size_t)N;
            float xa[RD1][R1], xb[RD1][R1];
#pragma unroll
            for (int i = 0; i < RD1; ++i) {
                const int q2 = 2 * (ad.g1[i] / (2 * N));  // first trace of this butterfly's pair within the unit
                const unsigned ga = (unsigned)(q2 < n_tr ? ad.g1[i] : ad.m1[i]);
                const unsigned gb = (unsigned)(q2 + 1 < n_tr ? ad.g1[i] + N : ad.m1[i]);
#pragma unroll
                for (int j1 = 0; j1 < R1; ++j1) {
                    xa[i][j1] = ld_off(x0, ga + (unsigned)(M1 * j1));
                    xb[i][j1] = ld_off(x0, gb + (unsigned)(M1 * j1));
                }
            }
#pragma unroll
            for (int i = 0; i < RD1; ++i) {
                const bool on = lane + kWave * i < Q * P::B1;
                const int m = ad.m1[i];
                const int q2 = 2 * (ad.g1[i] / (2 * N));
                const float sa = q2 < n_tr ? 1.0f : 0.0f, sb = q2 + 1 < n_tr ? 1.0f : 0.0f;
                cx v[R1];
#pragma unroll
                for (int j1 = 0; j1 < R1; ++j1) {
                    const float pw = pre_l[M1 * j1 + m];
                    v[j1] = cx{xa[i][j1] * (pw * sa), xb[i][j1] * (pw * sb)};
                }
                p_pass1_round<P>(v, buf, t1l, m, ad.l1[i], on);
                THZ_SCHED_FENCE();
            }
            p_pass23<P, Q>(buf, t2l, ad, lane);  // buf[q WE + k] = Z_q[k] = X1[k] + i X2[k] of pair q

            // ---- spectrum epilogue, pair by pair: X1 = (Z[k] + conj Z[N-k]) / 2, X2 = (Z[k] - conj Z[N-k]) / 2i (buf holds Z / 2)
#pragma unroll 1
            for (int q = 0; q < Q; ++q) {
                if (2 * q >= n_tr) break;  // wave-uniform
                const size_t p = p0 + 2 * (size_t)q;
                const bool has2 = 2 * q + 1 < n_tr;
                cx *bq = buf + q * WE;
                FBUnwrap u1, u2;
#pragma unroll 1
                for (int g = 0; g < n_groups; ++g) {
                    const int k0 = 256 * g + lb4;
                    cx X1[4], X2[4], h[4];
                    float m[4];
                    bool ok[4], rb[4];
                    int kcs[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int k = k0 + c;
                        ok[c] = k < NF;
                        const int kc = ok[c] ? k : NF - 1;
                        const int km = kc == 0 ? 0 : N - kc;  // Z[N] = Z[0]
                        kcs[c] = kc;
                        const cx Fk = bq[kc], Fm = bq[km];
                        X1[c] = cx{Fk.x + Fm.x, Fk.y - Fm.y};   // Z carries the factor 1/2 already (pre_s)
                        X2[c] = cx{Fk.y + Fm.y, Fm.x - Fk.x};
                        if constexpr (CM) h[c] = reinterpret_cast<const cx *>(mask_l)[kc];
                        else m[c] = mask_l[kc];
                        // real input: DC / Nyquist bins are real, with a POSITIVE zero as imaginary part (realfft writes
                        // +0.0 there; arg() of a negative real bin is then +pi, not -pi).  Forced on the bit pattern: a
                        // float select left -0.0 = -0.5 (x - x) standing in the 10 x 10 x 10 instantiation
                        {
                            const bool real_bin = kc == 0 || ((N & 1) == 0 && kc == NF - 1);
                            X1[c].y = p_zero_if(X1[c].y, real_bin);
                            X2[c].y = p_zero_if(X2[c].y, real_bin);
                            rb[c] = real_bin;
                        }
                    }
                    const size_t o1 = p * (size_t)NF + k0;
                    cx Y1[4], Y2[4];  // the multiplied spectra (CM)
                    if constexpr (CM) {
                        fb_finish_bins_c(X1, h, rb, ok, g, lane, u1, A.fft_out ? A.fft_out + o1 : nullptr,
                                         A.amp_out ? A.amp_out + o1 : nullptr, A.ph_out ? A.ph_out + o1 : nullptr, Y1);
                        if constexpr (SUMS) sums.group(g, 0, u1, true, lane);
                        if (has2)
                            fb_finish_bins_c(X2, h, rb, ok, g, lane, u2, A.fft_out ? A.fft_out + o1 + NF : nullptr,
                                             A.amp_out ? A.amp_out + o1 + NF : nullptr, A.ph_out ? A.ph_out + o1 + NF : nullptr, Y2);
                        else
                            for (int c = 0; c < 4; ++c) Y2[c] = cx{0.0f, 0.0f};
                    } else {
                        fb_finish_bins(X1, m, ok, g, lane, u1, A.fft_out ? A.fft_out + o1 : nullptr,
                                       A.amp_out ? A.amp_out + o1 : nullptr, A.ph_out ? A.ph_out + o1 : nullptr);
                        if constexpr (SUMS) sums.group(g, 0, u1, true, lane);
                        if (has2)
                            fb_finish_bins(X2, m, ok, g, lane, u2, A.fft_out ? A.fft_out + o1 + NF : nullptr,
                                           A.amp_out ? A.amp_out + o1 + NF : nullptr, A.ph_out ? A.ph_out + o1 + NF : nullptr);
                    }
                    if constexpr (SUMS) sums.group(g, 1, u2, has2, lane);
                    // input of the inverse, in place: conj(G[k]) and conj(G[N-k]) of G = Y1full + i Y2full
                    // (Yfull[n] = Y[n] up to N/2, conj(Y[N-n]) above) — bin k's owner is the only reader of
                    // slots k and N-k
                    if constexpr (MODE == kPipe) {
                        // no contraction here: y = X m must round before the sums below, so that the inverse
                        // transforms exactly the spectrum that was stored (a later Filter(6 / 7) update re-runs k_p<inv>
                        // on the stored one and has to land on the same samples)
#pragma clang fp contract(off)
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (ok[c]) {
                                // a missing second trace is exactly zero, as in k_p<inv> (its "spectrum" here is
                                // the rounding noise of the split, which would leak into the first trace's samples)
                                const cx y1 = CM ? Y1[c] : cx{X1[c].x * m[c], X1[c].y * m[c]};
                                const cx y2 = !has2 ? cx{0.0f, 0.0f} : CM ? Y2[c] : cx{X2[c].x * m[c], X2[c].y * m[c]};
                                const int kc = kcs[c];
                                bq[kc] = cx{y1.x - y2.y, -y1.y - y2.x};
                                if (kc != 0 && 2 * kc != N) bq[N - kc] = cx{y1.x + y2.y, y1.y - y2.x};
                            }
                    }
                }
            }
            wave_sync();
        } else {
            // inverse only: conj(G) from the two spectra in memory; DC (and Nyquist for even N) imaginary
            // parts are ignored like realfft's C2R does
#pragma unroll 1
            for (int q = 0; q < Q; ++q) {
                const bool live = 2 * q < n_tr, has2 = 2 * q + 1 < n_tr;
                const cx *f1 = A.fft_in + (p0 + (live ? 2 * (size_t)q : 0)) * (size_t)NF;
                cx *bq = buf + q * WE;
                // every bin's loads are issued before the first is used: one trip to HBM per pair (the loop form waited
                // for each pair of loads in turn — with the 7 to 12 waves of the longer plans that latency showed)
                constexpr int KI = (NF + kWave - 1) / kWave;
                cx y1v[KI], y2v[KI];
#pragma unroll
                for (int i = 0; i < KI; ++i) {
                    const int k = lb1 + kWave * i;
                    const unsigned kc = (unsigned)(k < NF ? k : NF - 1);  // in bounds; unused when k >= NF
                    y1v[i] = ld_off(f1, kc);
                    y2v[i] = ld_off(f1, has2 ? (unsigned)NF + kc : kc);
                }
#pragma unroll
                for (int i = 0; i < KI; ++i) {
                    const int k = lb1 + kWave * i;
                    if (k < NF) {
                        cx y1 = y1v[i];
                        cx y2 = has2 ? y2v[i] : cx{0.0f, 0.0f};
                        if (k == 0 || ((N & 1) == 0 && k == NF - 1)) {
                            y1.y = 0.0f;
                            y2.y = 0.0f;
                        }
                        bq[k] = cx{y1.x - y2.y, -y1.y - y2.x};
                        if (k != 0 && 2 * k != N) bq[N - k] = cx{y1.x + y2.y, y1.y - y2.x};
                    }
                }
            }
            wave_sync();
        }
        if constexpr (MODE == kFwd) continue;

        // ---- U = DFT(conj G): pass 1 in place from LDS, then passes 2 and 3
#pragma unroll
        for (int i = 0; i < RD1; ++i) {
            const bool on = lane + kWave * i < Q * P::B1;
            cx v[R1];
#pragma unroll
            for (int j1 = 0; j1 < R1; ++j1) v[j1] = buf[ad.l1[i] + M1 * j1];
            p_pass1_round<P>(v, buf, t1l, ad.m1[i], ad.l1[i], on);
            THZ_SCHED_FENCE();
        }
        p_pass23<P, Q>(buf, t2l, ad, lane);

        // ---- y1 = Re U / nt, y2 = -Im U / nt, each times post[n]; images = sum y^2
#pragma unroll 1
        for (int q = 0; q < Q; ++q) {
            if (2 * q >= n_tr) break;  // wave-uniform
            const size_t p = p0 + 2 * (size_t)q;
            const bool has2 = 2 * q + 1 < n_tr;
            const cx *bq = buf + q * WE;
            float *o1 = A.data_out + p * (size_t)N;
            float acc1 = 0.0f, acc2 = 0.0f;
            // four consecutive samples per lane: 16-byte stores (rows are only 4-byte aligned: store_f4)
            constexpr int QUADS = (N + 3) / 4, QR = (QUADS + kWave - 1) / kWave;
#pragma unroll
            for (int i = 0; i < QR; ++i) {
                const int n0 = lb4 + 4 * kWave * i;
                if (n0 + 3 < N) {
                    float v1[4], v2[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const cx U = bq[n0 + c];
                        const float pw = post_l[n0 + c];
                        v1[c] = by_nt(U.x) * pw;
                        v2[c] = by_nt(-U.y) * pw;
                        acc1 += v1[c] * v1[c];
                        acc2 += v2[c] * v2[c];
                    }
                    store_f4(o1 + n0, v1[0], v1[1], v1[2], v1[3]);
                    if (has2) store_f4(o1 + N + n0, v2[0], v2[1], v2[2], v2[3]);
                } else {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        if (n0 + c < N) {
                            const cx U = bq[n0 + c];
                            const float pw = post_l[n0 + c];
                            const float a = by_nt(U.x) * pw, b = by_nt(-U.y) * pw;
                            o1[n0 + c] = a;
                            acc1 += a * a;
                            if (has2) { o1[N + n0 + c] = b; acc2 += b * b; }
                        }
                }
            }
            if (!has2) acc2 = 0.0f;
            if (A.img) {
                acc1 = wave_reduce_add(acc1);
                acc2 = wave_reduce_add(acc2);
                if (lane == 0) {
                    A.img[p] = acc1;
                    if (has2) A.img[p + 1] = acc2;
                }
            }
        }
        wave_sync();
    }
    if constexpr (SUMS) sums.finish(A.sum_partial + (size_t)blockIdx.x * (size_t)(2 * NF), NF, (int)threadIdx.x, (int)blockDim.x);
}

}  // namespace thz
