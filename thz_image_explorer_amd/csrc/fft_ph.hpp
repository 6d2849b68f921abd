// fft_ph.hpp — "PH" kernels: EVEN trace lengths nt = 2 N whose half N = R1 R2 R3 is one of the P plans (fft_p.hpp):
// nt = 2002, 2400, 3000, 4000 (and 2000 on request).  realfft hands such a length to a complex transform of length N
// plus a split, like any even length (math_tools.rs:375, :549); the chirp-z kernels these lengths fell to before need
// four transforms of length 8192 per pair of traces (0.07 - 0.11 of the HBM roofline).
//
// One wave per TRACE, z[n] = x[2n] + i x[2n+1] (8-byte loads straight into the radix-R1 butterflies of pass 1), the
// mixed-radix core of the P kernels (p_pass1_round, p_pass23) on a buffer of N + 2 entries, then
//   split     X[k], X[N-k] from Z[k], Z[N-k] and W_2N^k, pair by pair, in place (slot N takes X[N])
//   finish    bins in ascending groups of 256: |X| m, arg X, numpy_unwrap, stores (fb_finish_bins — the epilogue of the
//             P and chirp-z kernels, one spectrum instead of two); the masked bins go back to the buffer
//   merge     conj(Z'[k]), conj(Z'[N-k]) from Y[k], Y[N-k] (realfft's C2R pre-processing, unnormalised), in place
//   inverse   U = DFT(conj Z'):  y[2n] = Re U[n] / nt, y[2n+1] = -Im U[n] / nt, times the post window; image = sum y^2
// Windows, mask and the split twiddles are read where they are used (L2): at N = 2000 a wave's buffer is 16 KB and
// eight of them plus the pass-1 table fill the CU's LDS.  No complex multiplier and no in-launch sums here: those
// requests take the entry points' general paths (thz_apply_fd_cmask, thz_pixel_sum).
#pragma once

#include "fft_p.hpp"

namespace thz {

template <class P>
struct PHLayout {
    static constexpr int N = P::N;
    static constexpr int WE = N + 2;  // Z[0 .. N-1], X[N], one spare (16-byte rows)
    static constexpr size_t lds_bytes(int waves) { return (size_t)(P::T1_ENTRIES + P::T2_ENTRIES + waves * WE) * sizeof(cx); }
    // waves of a block: what LDS holds, 13 to 15 become 12 (p_block_waves), and no more than the registers of the
    // plan's widest butterfly allow without spilling (radix 20: two waves per SIMD, radix 15: three)
    static constexpr int waves()
    {
        int w = 16;
        while (w > 1 && lds_bytes(w) > (size_t)160 * 1024) --w;
        if (w >= 13 && w < 16) w = 12;
        const int cap = P::R3 >= 20 ? 8 : P::R3 >= 15 ? 12 : 16;
        return w < cap ? w : cap;
    }
};

struct PHTables {
    const cx *t1;  // [k1][m]  W_N^(m k1)
    const cx *t2;  // [k2][j3] W_(R2 R3)^(j3 k2)
    const cx *w2;  // W_2N^k = exp(-i pi k / N), k <= N / 2
};

template <class P, int MODE>
__global__ __launch_bounds__(PHLayout<P>::waves() * kWave) void k_ph(FBArgs A, PHTables T)
{
    THZ_DYN_LDS(lds);
    constexpr int N = P::N, NT = 2 * N, NF = N + 1, R1 = P::R1, M1 = P::M1, WE = PHLayout<P>::WE;
    constexpr int RD1 = PAddr<P, 1>::RD1;
    constexpr int HP = N / 2 + 1;                      // pairs (k, N - k), k = 0 .. N / 2
    constexpr int KP = (HP + kWave - 1) / kWave;       // ... per lane
    const int lane = lane_id();
    const int wib = THZ_UNIFORM((int)(threadIdx.x >> 6));
    const int wpb = (int)(blockDim.x >> 6);
    cx *t1 = reinterpret_cast<cx *>(lds);
    cx *t2 = t1 + P::T1_ENTRIES;
    cx *buf = t2 + P::T2_ENTRIES + (size_t)wib * WE;
    for (int i = (int)threadIdx.x; i < P::T1_ENTRIES; i += (int)blockDim.x) t1[i] = T.t1[i];
    for (int i = (int)threadIdx.x; i < M1; i += (int)blockDim.x) t2[i] = T.t2[i];
    __syncthreads();

    PAddr<P, 1> ad;
    ad.init(lane);
    const DivConst by_nt((float)NT);
    constexpr int n_groups = (NF + 255) / 256;
    const size_t stride = (size_t)gridDim.x * wpb;

    for (size_t p = (size_t)blockIdx.x * wpb + wib; p < A.npix; p += stride) {
        ad.refresh();
        const cx *t1l = launder_uniform((const cx *)t1);
        const cx *t2l = launder_uniform((const cx *)t2);
        const cx *w2 = launder_uniform(T.w2);
        const float *mask_g = launder_uniform(A.mask);
        const float *pre_g = A.pre_win ? launder_uniform(A.pre_win) : nullptr;
        const float *post_g = A.post_win ? launder_uniform(A.post_win) : nullptr;
        const int lb4 = launder_v(4 * lane), lb1 = launder_v(lane);

        if constexpr (MODE != kInv) {
            // ---- pass 1 from memory: z[n] = (x[2n], x[2n+1]) pre, n = M1 j1 + m; every round's loads are issued before
            // the first butterfly (one trip to HBM per trace).  The 1/2 of the split rides on the window (exact).
            const float2 *x0 = reinterpret_cast<const float2 *>(A.in + p * (size_t)NT);
            float2 xa[RD1][R1];
#pragma unroll
            for (int i = 0; i < RD1; ++i)
#pragma unroll
                for (int j1 = 0; j1 < R1; ++j1) xa[i][j1] = ld_off(x0, (unsigned)(ad.m1[i] + M1 * j1));
#pragma unroll
            for (int i = 0; i < RD1; ++i) {
                const bool on = lane + kWave * i < P::B1;
                const int m = ad.m1[i];
                cx v[R1];
#pragma unroll
                for (int j1 = 0; j1 < R1; ++j1) {
                    float2 pw = make_float2(1.0f, 1.0f);
                    if (pre_g) pw = ld_off(reinterpret_cast<const float2 *>(pre_g), (unsigned)(M1 * j1 + m));
                    v[j1] = cx{xa[i][j1].x * (0.5f * pw.x), xa[i][j1].y * (0.5f * pw.y)};
                }
                p_pass1_round<P>(v, buf, t1l, m, ad.l1[i], on);
                THZ_SCHED_FENCE();
            }
            p_pass23<P, 1>(buf, t2l, ad, lane);  // buf[k] = Z[k] / 2

            // ---- split, pair by pair:  E = Z[k] + conj Z[N-k],  O = (Z[k] - conj Z[N-k]) / i  (the halves are in Z),
            // X[k] = E + w O,  X[N-k] = conj(E - w O),  w = W_2N^k
            {
                cx wv[KP];
#pragma unroll
                for (int i = 0; i < KP; ++i) {
                    const int k = lb1 + kWave * i;
                    wv[i] = ld_off(w2, (unsigned)(k < HP ? k : 0));
                }
#pragma unroll
                for (int i = 0; i < KP; ++i) {
                    const int k = lb1 + kWave * i;
                    if (k < HP) {
                        const cx a = buf[k], b = buf[k == 0 ? 0 : N - k];
                        const cx E = cx{a.x + b.x, a.y - b.y};
                        const cx O = cx{a.y + b.y, b.x - a.x};
                        const cx wo = cx_mul(wv[i], O);
                        cx xk = cx{E.x + wo.x, E.y + wo.y};
                        cx xn = cx{E.x - wo.x, wo.y - E.y};
                        if (k == 0) {  // real bins, with a positive zero as imaginary part (realfft writes +0.0)
                            xk.y = p_zero_if(xk.y, true);
                            xn.y = p_zero_if(xn.y, true);
                        }
                        buf[k] = xk;
                        buf[N - k] = xn;  // k = 0: slot N takes X[N]
                    }
                }
            }
            wave_sync();

            // ---- finish: ascending groups of 256 bins (the unwrap's carry runs over them)
            {
                FBUnwrap u;
#pragma unroll 1
                for (int g = 0; g < n_groups; ++g) {
                    const int k0 = 256 * g + lb4;
                    cx X[4];
                    float m[4];
                    bool ok[4];
                    int kcs[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int k = k0 + c;
                        ok[c] = k < NF;
                        kcs[c] = ok[c] ? k : NF - 1;
                        X[c] = buf[kcs[c]];
                        m[c] = ld_off(mask_g, (unsigned)kcs[c]);
                    }
                    const size_t o = p * (size_t)NF + k0;
                    fb_finish_bins(X, m, ok, g, lane, u, A.fft_out ? A.fft_out + o : nullptr, A.amp_out ? A.amp_out + o : nullptr,
                                   A.ph_out ? A.ph_out + o : nullptr);
                    if constexpr (MODE == kPipe) {
                        // the inverse transforms exactly the spectrum that was stored: the product rounds here, alone
#pragma clang fp contract(off)
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (ok[c]) buf[kcs[c]] = cx{X[c].x * m[c], X[c].y * m[c]};
                    }
                }
            }
            wave_sync();
        }
        if constexpr (MODE == kFwd) continue;

        // ---- merge: conj Z'[k] = conj(A) - i conj(B), conj Z'[N-k] = A - i B with A = Y[k] + conj Y[N-k],
        // B = (Y[k] - conj Y[N-k]) conj(w); the imaginary parts of Y[0] and Y[N] are ignored like realfft's C2R does
        {
            const cx *f1 = MODE == kInv ? A.fft_in + p * (size_t)NF : nullptr;
            cx wv[KP], yk[KP], yn[KP];
#pragma unroll
            for (int i = 0; i < KP; ++i) {
                const int k = lb1 + kWave * i;
                const int kc = k < HP ? k : 0;
                wv[i] = ld_off(w2, (unsigned)kc);
                if constexpr (MODE == kInv) {
                    yk[i] = ld_off(f1, (unsigned)kc);
                    yn[i] = ld_off(f1, (unsigned)(N - kc));
                }
            }
#pragma unroll
            for (int i = 0; i < KP; ++i) {
                const int k = lb1 + kWave * i;
                if (k < HP) {
                    cx a = MODE == kInv ? yk[i] : buf[k];
                    cx b = MODE == kInv ? yn[i] : buf[N - k];
                    if (k == 0) {
                        a.y = 0.0f;
                        b.y = 0.0f;
                    }
                    const cx Aa = cx{a.x + b.x, a.y - b.y};
                    const cx D = cx{a.x - b.x, a.y + b.y};
                    const cx B = cx_mul(D, cx{wv[i].x, -wv[i].y});
                    buf[k] = cx{Aa.x - B.y, -Aa.y - B.x};          // conj(A) - i conj(B)
                    if (k != 0) buf[N - k] = cx{Aa.x + B.y, Aa.y - B.x};  // A - i B  (N even, k = N / 2: the same value)
                }
            }
        }
        wave_sync();

        // ---- U = DFT(conj Z'): pass 1 in place from LDS, then passes 2 and 3
#pragma unroll
        for (int i = 0; i < RD1; ++i) {
            const bool on = lane + kWave * i < P::B1;
            cx v[R1];
#pragma unroll
            for (int j1 = 0; j1 < R1; ++j1) v[j1] = buf[ad.l1[i] + M1 * j1];
            p_pass1_round<P>(v, buf, t1l, ad.m1[i], ad.l1[i], on);
            THZ_SCHED_FENCE();
        }
        p_pass23<P, 1>(buf, t2l, ad, lane);

        // ---- y[2n] = Re U[n] / nt, y[2n+1] = -Im U[n] / nt, each times post; image = sum y^2.  Four consecutive samples
        // (two entries) per lane: 16-byte stores on rows that are 8-byte aligned (nt is even)
        {
            float *o1 = A.data_out + p * (size_t)NT;
            float acc = 0.0f;
            constexpr int QUADS = (NT + 3) / 4, QR = (QUADS + kWave - 1) / kWave;
#pragma unroll
            for (int i = 0; i < QR; ++i) {
                const int s0 = lb4 + 4 * kWave * i;  // first sample of the quad
                if (s0 + 3 < NT) {
                    const cx U0 = buf[s0 / 2], U1 = buf[s0 / 2 + 1];
                    float4 pw = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
                    if (post_g) {
                        const float2 pa = ld_off(reinterpret_cast<const float2 *>(post_g), (unsigned)(s0 / 2));
                        const float2 pb = ld_off(reinterpret_cast<const float2 *>(post_g), (unsigned)(s0 / 2 + 1));
                        pw = make_float4(pa.x, pa.y, pb.x, pb.y);
                    }
                    const float v0 = by_nt(U0.x) * pw.x, v1 = by_nt(-U0.y) * pw.y, v2 = by_nt(U1.x) * pw.z, v3 = by_nt(-U1.y) * pw.w;
                    acc += v0 * v0;
                    acc += v1 * v1;
                    acc += v2 * v2;
                    acc += v3 * v3;
                    store_f4(o1 + s0, v0, v1, v2, v3);
                } else if (s0 < NT) {  // nt = 2 N with N odd: one entry left
                    const cx U0 = buf[s0 / 2];
                    const float pa = post_g ? ld_off(post_g, (unsigned)s0) : 1.0f, pb = post_g ? ld_off(post_g, (unsigned)(s0 + 1)) : 1.0f;
                    const float v0 = by_nt(U0.x) * pa, v1 = by_nt(-U0.y) * pb;
                    acc += v0 * v0;
                    acc += v1 * v1;
                    o1[s0] = v0;
                    o1[s0 + 1] = v1;
                }
            }
            if (A.img) {
                acc = wave_reduce_add(acc);
                if (lane == 0) A.img[p] = acc;
            }
        }
        wave_sync();
    }
}

}  // namespace thz
