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
    // LDS per block: [T1: N cx, [k1][m]][T2: R2 R3 cx, [k2][j3]][mask nf][pre N][post N] floats, [per wave: N cx (+pad)]
    static constexpr int T1_ENTRIES = N, T2_ENTRIES = M1 + (M1 & 1);
    static constexpr int WAVE_ENTRIES = N + (N & 1);
    static constexpr size_t lds_bytes(int waves)
    {
        return (size_t)(T1_ENTRIES + T2_ENTRIES + waves * WAVE_ENTRIES) * sizeof(cx)
               + (size_t)(pad4(NF) + 2 * pad4(N)) * sizeof(float);
    }
};

using PPlan1001 = PPlan<7, 11, 13>;
using PPlan1000 = PPlan<10, 10, 10>;

struct PTables {
    const cx *t1;  // [k1][m]  W_N^(m k1),           N entries
    const cx *t2;  // [k2][j3] W_(R2 R3)^(j3 k2),    R2 R3 entries
};

// lane -> butterfly maps of the three passes (trace-invariant; kept per round as LDS base indices)
template <class P>
struct PAddr {
    int b1[P::RD1];  // pass 1: m            (butterfly m = lane + 64 round), clamped
    int b2[P::RD2];  // pass 2: k1 M1 + j3   (butterfly b = k1 R3 + j3)
    int t2[P::RD2];  // pass 2: j3
    int b3[P::RD3];  // pass 3: k1 M1 + R3 k2 (butterfly b = k1 + R1 k2)
    int o3[P::RD3];  // pass 3: b
    // A lane without a butterfly in a pass's last round works on the butterfly it already finished in round 0
    // (its own, so nobody else is writing there) and does not store: no lane ever reads what another lane writes
    // in the same round — true on the GPU by lock-step anyway, and what keeps the emulation ThreadSanitizer-clean.
    __device__ __forceinline__ void init(int lane)
    {
        static_assert(P::B1 >= kWave && P::B2 >= kWave && P::B3 >= kWave, "round 0 is full in every pass");
#pragma unroll
        for (int i = 0; i < P::RD1; ++i) {
            const int b = lane + kWave * i;
            b1[i] = b < P::B1 ? b : lane;
        }
#pragma unroll
        for (int i = 0; i < P::RD2; ++i) {
            int b = lane + kWave * i;
            b = b < P::B2 ? b : lane;
            b2[i] = (b / P::R3) * P::M1 + b % P::R3;
            t2[i] = b % P::R3;
        }
#pragma unroll
        for (int i = 0; i < P::RD3; ++i) {
            int b = lane + kWave * i;
            b = b < P::B3 ? b : lane;
            b3[i] = (b % P::R1) * P::M1 + P::R3 * (b / P::R1);
            o3[i] = b;
        }
    }
    __device__ __forceinline__ void refresh()
    {
#pragma unroll
        for (int i = 0; i < P::RD1; ++i) b1[i] = launder_v(b1[i]);
#pragma unroll
        for (int i = 0; i < P::RD2; ++i) { b2[i] = launder_v(b2[i]); t2[i] = launder_v(t2[i]); }
#pragma unroll
        for (int i = 0; i < P::RD3; ++i) { b3[i] = launder_v(b3[i]); o3[i] = launder_v(o3[i]); }
    }
};

// pass 1 of one round: v[j1] holds z[M1 j1 + m]; writes y[k1][m] W_N^(m k1) to buf[k1 M1 + m]
template <class P>
__device__ __forceinline__ void p_pass1_round(cx (&v)[P::R1], cx *buf, const cx *t1, int m, bool on)
{
    PDft<P::R1>::run(v);
    if (on) {
        buf[m] = v[0];
#pragma unroll
        for (int k1 = 1; k1 < P::R1; ++k1) buf[k1 * P::M1 + m] = cx_mul(v[k1], t1[k1 * P::M1 + m]);
    }
}

// passes 2 and 3 on buf; leaves the natural-order transform in buf[0 .. N).  Ends with wave_sync().
template <class P>
__device__ __forceinline__ void p_pass23(cx *buf, const cx *t2, const PAddr<P> &ad, int lane)
{
    constexpr int R1 = P::R1, R2 = P::R2, R3 = P::R3;
    wave_sync();
    // ---- pass 2: butterfly (k1, j3), elements k1 M1 + R3 j2 + j3, in place
#pragma unroll
    for (int i = 0; i < P::RD2; ++i) {
        const bool on = lane + kWave * i < P::B2;
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
    cx d[P::RD3][R3];
#pragma unroll
    for (int i = 0; i < P::RD3; ++i) {
#pragma unroll
        for (int j3 = 0; j3 < R3; ++j3) d[i][j3] = buf[ad.b3[i] + j3];
    }
    wave_sync();
#pragma unroll
    for (int i = 0; i < P::RD3; ++i) {
        const bool on = lane + kWave * i < P::B3;
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

// Same argument block as the chirp-z kernels (w / bf unused)
template <class P, int MODE>
__global__ __launch_bounds__(1024) void k_p(FBArgs A, PTables T)
{
    THZ_DYN_LDS(lds);
    constexpr int N = P::N, NF = P::NF, R1 = P::R1, M1 = P::M1;
    const int lane = lane_id();
    const int wib = THZ_UNIFORM((int)(threadIdx.x >> 6));
    const int wpb = (int)(blockDim.x >> 6);
    cx *t1 = reinterpret_cast<cx *>(lds);
    cx *t2 = t1 + P::T1_ENTRIES;
    cx *buf = t2 + P::T2_ENTRIES + (size_t)wib * P::WAVE_ENTRIES;
    float *mask_s = reinterpret_cast<float *>(t2 + P::T2_ENTRIES + (size_t)wpb * P::WAVE_ENTRIES);
    float *pre_s = mask_s + P::pad4(NF);
    float *post_s = pre_s + P::pad4(N);
    for (int i = (int)threadIdx.x; i < P::T1_ENTRIES; i += (int)blockDim.x) t1[i] = T.t1[i];
    for (int i = (int)threadIdx.x; i < M1; i += (int)blockDim.x) t2[i] = T.t2[i];
    for (int i = (int)threadIdx.x; i < NF; i += (int)blockDim.x) mask_s[i] = A.mask[i];
    for (int i = (int)threadIdx.x; i < N; i += (int)blockDim.x) {
        pre_s[i] = A.pre_win ? A.pre_win[i] : 1.0f;
        post_s[i] = A.post_win ? A.post_win[i] : 1.0f;
    }
    __syncthreads();

    PAddr<P> ad;
    ad.init(lane);
    const float fnt = (float)N;
    constexpr int n_groups = (NF + 255) / 256;  // epilogue groups of 256 bins: bin = 256 g + 4 lane + c
    const size_t n_pairs = (A.npix + 1) / 2;
    const size_t stride = (size_t)gridDim.x * wpb;

    for (size_t q = (size_t)blockIdx.x * wpb + wib; q < n_pairs; q += stride) {
        const size_t p = 2 * q;
        const bool has2 = p + 1 < A.npix;  // wave-uniform
        ad.refresh();
        const cx *t1l = launder_uniform((const cx *)t1);
        const cx *t2l = launder_uniform((const cx *)t2);
        const float *pre_l = launder_uniform((const float *)pre_s);
        const float *post_l = launder_uniform((const float *)post_s);
        const float *mask_l = launder_uniform((const float *)mask_s);
        const int lb4 = launder_v(4 * lane), lb1 = launder_v(lane);

        if constexpr (MODE != kInv) {
            // ---- pass 1 from memory: z[n] = (x1[n] + i x2[n]) pre[n], n = M1 j1 + m; a round's loads
            // are issued together (clamped indices, predicated stores)
            const float *x1 = A.in + p * (size_t)N;
            const float *x2 = has2 ? x1 + N : x1;
            // every round's loads are issued before the first butterfly: one trip to HBM per pair, not one per round
            float xa[P::RD1][R1], xb[P::RD1][R1];
#pragma unroll
            for (int i = 0; i < P::RD1; ++i) {
#pragma unroll
                for (int j1 = 0; j1 < R1; ++j1) {
                    xa[i][j1] = ld_off(x1, (unsigned)(M1 * j1 + ad.b1[i]));
                    xb[i][j1] = ld_off(x2, (unsigned)(M1 * j1 + ad.b1[i]));
                }
            }
#pragma unroll
            for (int i = 0; i < P::RD1; ++i) {
                const bool on = lane + kWave * i < P::B1;
                const int m = ad.b1[i];
                cx v[R1];
#pragma unroll
                for (int j1 = 0; j1 < R1; ++j1) {
                    const float pw = pre_l[M1 * j1 + m];
                    v[j1] = cx{xa[i][j1] * pw, has2 ? xb[i][j1] * pw : 0.0f};
                }
                p_pass1_round<P>(v, buf, t1l, m, on);
                THZ_SCHED_FENCE();
            }
            p_pass23<P>(buf, t2l, ad, lane);  // buf[k] = Z[k] = X1[k] + i X2[k]

            // ---- spectrum epilogue: X1 = (Z[k] + conj Z[N-k]) / 2, X2 = (Z[k] - conj Z[N-k]) / 2i
            {
                FBUnwrap u1, u2;
#pragma unroll 1
                for (int g = 0; g < n_groups; ++g) {
                    const int k0 = 256 * g + lb4;
                    cx X1[4], X2[4];
                    float m[4];
                    bool ok[4];
                    int kcs[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int k = k0 + c;
                        ok[c] = k < NF;
                        const int kc = ok[c] ? k : NF - 1;
                        const int km = kc == 0 ? 0 : N - kc;  // Z[N] = Z[0]
                        kcs[c] = kc;
                        const cx Fk = buf[kc], Fm = buf[km];
                        X1[c] = cx{0.5f * (Fk.x + Fm.x), 0.5f * (Fk.y - Fm.y)};
                        X2[c] = cx{0.5f * (Fk.y + Fm.y), -0.5f * (Fk.x - Fm.x)};
                        m[c] = mask_l[kc];
                        // real input: DC / Nyquist bins are real, with a POSITIVE zero as imaginary part (realfft writes
                        // +0.0 there; arg() of a negative real bin is then +pi, not -pi).  Forced on the bit pattern: a
                        // float select left -0.0 = -0.5 (x - x) standing in the 10 x 10 x 10 instantiation
                        {
                            const bool real_bin = kc == 0 || ((N & 1) == 0 && kc == NF - 1);
                            X1[c].y = p_zero_if(X1[c].y, real_bin);
                            X2[c].y = p_zero_if(X2[c].y, real_bin);
                        }
                    }
                    const size_t o1 = p * (size_t)NF + k0;
                    fb_finish_bins(X1, m, ok, g, lane, u1, A.fft_out ? A.fft_out + o1 : nullptr,
                                   A.amp_out ? A.amp_out + o1 : nullptr, A.ph_out ? A.ph_out + o1 : nullptr);
                    if (has2)
                        fb_finish_bins(X2, m, ok, g, lane, u2, A.fft_out ? A.fft_out + o1 + NF : nullptr,
                                       A.amp_out ? A.amp_out + o1 + NF : nullptr, A.ph_out ? A.ph_out + o1 + NF : nullptr);
                    // input of the inverse, in place: conj(G[k]) and conj(G[N-k]) of G = Y1full + i Y2full
                    // (Yfull[n] = Y[n] up to N/2, conj(Y[N-n]) above) — bin k's owner is the only reader of
                    // slots k and N-k
                    if constexpr (MODE == kPipe) {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (ok[c]) {
                                const cx y1 = cx{X1[c].x * m[c], X1[c].y * m[c]}, y2 = cx{X2[c].x * m[c], X2[c].y * m[c]};
                                const int kc = kcs[c];
                                buf[kc] = cx{y1.x - y2.y, -y1.y - y2.x};
                                if (kc != 0 && 2 * kc != N) buf[N - kc] = cx{y1.x + y2.y, y1.y - y2.x};
                            }
                    }
                }
            }
            wave_sync();
        } else {
            // inverse only: conj(G) from the two spectra in memory; DC (and Nyquist for even N) imaginary
            // parts are ignored like realfft's C2R does
            const cx *f1 = A.fft_in + p * (size_t)NF;
            for (int k = lb1; k < NF; k += kWave) {
                cx y1 = ld_off(f1, (unsigned)k);
                cx y2 = has2 ? ld_off(f1, (unsigned)(NF + k)) : cx{0.0f, 0.0f};
                if (k == 0 || ((N & 1) == 0 && k == NF - 1)) {
                    y1.y = 0.0f;
                    y2.y = 0.0f;
                }
                buf[k] = cx{y1.x - y2.y, -y1.y - y2.x};
                if (k != 0 && 2 * k != N) buf[N - k] = cx{y1.x + y2.y, y1.y - y2.x};
            }
            wave_sync();
        }
        if constexpr (MODE == kFwd) continue;

        // ---- U = DFT(conj G): pass 1 in place from LDS, then passes 2 and 3
#pragma unroll
        for (int i = 0; i < P::RD1; ++i) {
            const bool on = lane + kWave * i < P::B1;
            const int m = ad.b1[i];
            cx v[R1];
#pragma unroll
            for (int j1 = 0; j1 < R1; ++j1) v[j1] = buf[M1 * j1 + m];
            p_pass1_round<P>(v, buf, t1l, m, on);
            THZ_SCHED_FENCE();
        }
        p_pass23<P>(buf, t2l, ad, lane);

        // ---- y1 = Re U / nt, y2 = -Im U / nt, each times post[n]; images = sum y^2
        {
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
                        const cx U = buf[n0 + c];
                        const float pw = post_l[n0 + c];
                        v1[c] = (U.x / fnt) * pw;
                        v2[c] = (-U.y / fnt) * pw;
                        acc1 += v1[c] * v1[c];
                        acc2 += v2[c] * v2[c];
                    }
                    store_f4(o1 + n0, v1[0], v1[1], v1[2], v1[3]);
                    if (has2) store_f4(o1 + N + n0, v2[0], v2[1], v2[2], v2[3]);
                } else {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        if (n0 + c < N) {
                            const cx U = buf[n0 + c];
                            const float pw = post_l[n0 + c];
                            const float a = (U.x / fnt) * pw, b = (-U.y / fnt) * pw;
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
}

}  // namespace thz
