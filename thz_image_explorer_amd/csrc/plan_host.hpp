// plan_host.hpp — host-side construction of the transform tables ("plan").
// Replaces RealFftPlanner::plan_fft_forward/plan_fft_inverse (io.rs:616-618,
// data_thread.rs:1200-1202, tilt_compensation.rs:210-212): what those calls
// prepare inside realfft/rustfft is here a set of twiddle tables computed in
// double precision and rounded once to f32.
#pragma once

#include "kernels.hpp"

#include <cmath>
#include <complex>
#include <vector>

namespace thz {

struct PlanHost {
    int nt = 0, nf = 0, mode = kModePow2, log2n = 0;
    int buf_entries = 0, lds_per_wave = 0, waves_per_block = 1;
    std::vector<c32> tw, tw_split, chirp_conj, bfft;
    int family = kFamilyG;
    std::vector<c32> f_t1, f_t2, f_w2n;
    std::vector<c32> p_t1, p_t2;  // P family (PH: of the half length, the split twiddles behind p_t2)
    int half_n = 0;                // PH kernels: nt = 2 half_n
    bool big = false;              // buffers in global scratch (k_fft_fwd_big / k_fft_inv_big)
    int big_waves = 0;             // waves of their grid = slots of the scratch
    const char *variant = "";
};

// P-family factorisation nt = r1 r2 r3 (fft_p.hpp): the lengths a kernel is instantiated for
inline bool p_factors(size_t nt, int &r1, int &r2, int &r3)
{
    switch (nt) {
    case 1001: r1 = 7; r2 = 11; r3 = 13; return true;   // every real scan of the reference's sample data
    case 1000: r1 = 10; r2 = 10; r3 = 10; return true;
    case 1200: r1 = 10; r2 = 10; r3 = 12; return true;  // round lengths of other instruments: same kernels,
    case 1500: r1 = 10; r2 = 10; r3 = 15; return true;  // one pair of traces per wave only
    case 2000: r1 = 10; r2 = 10; r3 = 20; return true;
    default: return false;
    }
}

// F-family factorisation of the half-length complex transform (fft_f.hpp)
inline bool f_factors(size_t nt, int &r1, int &r2, int &r3)
{
    switch (nt) {
    case 4096: r1 = 16; r2 = 16; r3 = 8; return true;
    case 2048: r1 = 8; r2 = 16; r3 = 8; return true;
    case 1024: r1 = 8; r2 = 8; r3 = 8; return true;
    default: return false;
    }
}

// Core tables of the 512-point complex transform (FPlan1024 = 8 x 8 x 8) on their own: the edge transform of the
// deconvolution's Parseval band energies (k_dc_energy_pv) — the same formulas as build_plan's F tables below
inline void dc_pv_core_tables(std::vector<c32> &t1, std::vector<c32> &t2)
{
    const double pi = 3.14159265358979323846;
    const size_t Nc = 512, m1 = 64;
    t1.resize(8 * m1);
    for (size_t k1 = 0; k1 < 8; ++k1)
        for (size_t m = 0; m < m1; ++m) {
            const double a = -2.0 * pi * (double)((m * k1) % Nc) / (double)Nc;
            t1[k1 * m1 + m] = c32{(float)std::cos(a), (float)std::sin(a)};
        }
    t2.resize(64);
    for (int k2 = 0; k2 < 8; ++k2)
        for (int j3 = 0; j3 < 8; ++j3) {
            const double a = -2.0 * pi * (double)((j3 * k2) % 64) / 64.0;
            t2[(size_t)k2 * 8 + j3] = c32{(float)std::cos(a), (float)std::sin(a)};
        }
}

inline bool is_pow2(size_t v) { return v && !(v & (v - 1)); }

// in-place iterative radix-2 FFT in double (host, table construction only)
inline void host_fft_pow2(std::vector<std::complex<double>> &a)
{
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    const double pi = 3.14159265358979323846;
    for (size_t len = 2; len <= n; len <<= 1) {
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const double ang = -2.0 * pi * (double)k / (double)len;
                const std::complex<double> w(std::cos(ang), std::sin(ang));
                const std::complex<double> u = a[i + k], v = a[i + k + len / 2] * w;
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
            }
    }
}

// Returns false when nt is outside the supported range.
inline bool build_plan(size_t nt, PlanHost &P, bool allow_f = true, bool allow_p = true)
{
    const double pi = 3.14159265358979323846;
    if (nt < 2) return false;
    P.nt = (int)nt;
    P.nf = (int)(nt / 2 + 1);
    size_t N;
    // Lengths whose buffers do not fit the CU's LDS run the G kernels on global scratch (round 3; realfft plans any
    // length, io.rs:616-618): powers of two above 16384, anything else above 8191 (4096 without the F core's
    // kernels) — up to kMaxTraceLength, where a wave's two buffers are 2 MB.
    constexpr size_t kMaxTraceLength = 65536;
    if (nt > kMaxTraceLength) return false;
    P.big = false;
    if (is_pow2(nt) && nt >= 4) {
        P.big = nt > 16384;
        P.mode = kModePow2;
        N = nt / 2;
        P.variant = P.big ? "g-stockham-global-scratch-r4r2" : "g-stockham-lds-r4r2";
    } else {
        // chirp-z: the G kernels hold 2 M entries per wave in LDS (nt <= 4096); above that the
        // FBS kernels (eight F-core runs per transform) up to nt = 8191
        P.big = nt > (allow_f ? 8191u : 4096u);
        P.mode = kModeBluestein;
        N = 1;
        while (N < 2 * nt - 1) N <<= 1;
        if (allow_f && N < 512) N = 512;  // smallest convolution length of the FB kernels (fft_fb.hpp)
        P.variant = P.big ? "g-bluestein-stockham-global-scratch-r4r2" : "g-bluestein-stockham-lds-r4r2";
    }
    int lg = 0;
    while (((size_t)1 << lg) < N) ++lg;
    P.log2n = lg;
    P.buf_entries = (int)(((N > (size_t)P.nf ? N : (size_t)P.nf) + 1) & ~(size_t)1);
    P.lds_per_wave = 2 * P.buf_entries * (int)sizeof(c32);
    int wpb = (int)(kLdsBytesPerCU / (size_t)P.lds_per_wave);
    P.big_waves = 0;
    if (P.big) {
        // as many waves as 1 GiB of scratch holds (a pass of a wave is a chain of dependent memory round trips: the
        // chip wants thousands of them in flight), 256 at least, in blocks of four
        size_t w = ((size_t)1 << 30) / (size_t)P.lds_per_wave;
        if (w > 4096) w = 4096;
        if (w < 256) w = 256;
        P.big_waves = (int)(w & ~(size_t)3);
        wpb = 4;
    } else if (wpb < 1) {
        if (!(allow_f && P.mode == kModeBluestein && N == 16384)) return false;
        wpb = 1;  // no G kernel can run this plan; every entry point goes through k_fbs<8>
    }
    if (wpb > 4) wpb = 4;
    P.waves_per_block = wpb;

    P.tw.resize(N);
    for (size_t m = 0; m < N; ++m) {
        const double a = -2.0 * pi * (double)m / (double)N;
        P.tw[m] = c32{(float)std::cos(a), (float)std::sin(a)};
    }
    P.tw_split.clear();
    P.chirp_conj.clear();
    P.bfft.clear();
    if (P.mode == kModePow2) {
        P.tw_split.resize(N / 2 + 1);
        for (size_t k = 0; k <= N / 2; ++k) {
            const double a = -2.0 * pi * (double)k / (double)nt;
            P.tw_split[k] = c32{(float)std::cos(a), (float)std::sin(a)};
        }
    } else {
        // chirp[n] = exp(+i*pi*n^2/nt); reduce n^2 mod 2nt in integers first
        std::vector<std::complex<double>> chirp(nt);
        for (size_t n = 0; n < nt; ++n) {
            const unsigned long long q = ((unsigned long long)n * n) % (2ull * nt);
            const double a = pi * (double)q / (double)nt;
            chirp[n] = std::complex<double>(std::cos(a), std::sin(a));
        }
        P.chirp_conj.resize(nt);
        for (size_t n = 0; n < nt; ++n)
            P.chirp_conj[n] = c32{(float)chirp[n].real(), (float)(-chirp[n].imag())};
        std::vector<std::complex<double>> b(N, std::complex<double>(0.0, 0.0));
        b[0] = chirp[0];
        for (size_t n = 1; n < nt; ++n) {
            b[n] = chirp[n];
            b[N - n] = chirp[n];
        }
        host_fft_pow2(b);
        P.bfft.resize(N);
        for (size_t m = 0; m < N; ++m)
            P.bfft[m] = c32{(float)(b[m].real() / (double)N), (float)(b[m].imag() / (double)N)};
    }
    P.family = kFamilyG;
    P.f_t1.clear(); P.f_t2.clear(); P.f_w2n.clear();
    int r1, r2, r3;
    // F core of complex length Nc: the half-length transform of a power-of-two trace (family F), or
    // the length-M convolution transform of a chirp-z length with M <= 2048 (family FB)
    // ... or two core runs per length-4096 convolution (family FB2, 1024 < nt < 2048)
    const bool fb8 = allow_f && !P.big && P.mode == kModeBluestein && N == 16384 && f_factors(N / 4, r1, r2, r3);
    const bool fb4 = allow_f && P.mode == kModeBluestein && N == 8192 && f_factors(N / 2, r1, r2, r3);
    const bool fb2 = allow_f && P.mode == kModeBluestein && N == 4096 && f_factors(N, r1, r2, r3);
    const bool fb = !fb2 && allow_f && P.mode == kModeBluestein && f_factors(2 * N, r1, r2, r3);
    if (fb || fb2 || fb4 || fb8 || (allow_f && P.mode == kModePow2 && f_factors(nt, r1, r2, r3))) {
        P.family = fb8 ? kFamilyFB8 : fb4 ? kFamilyFB4 : fb2 ? kFamilyFB2 : (fb ? kFamilyFB : kFamilyF);
        P.variant = fb8 ? "fb8-bluestein-regs-3pass-lds-xor" : fb4 ? "fb4-bluestein-regs-3pass-lds-xor" : fb2 ? "fb2-bluestein-regs-3pass-lds-xor" : (fb ? "fb-bluestein-regs-3pass-lds-xor" : "f-regs-3pass-lds-xor");
        const size_t Nc = fb8 ? N / 8 : fb4 ? N / 4 : fb2 ? N / 2 : (fb ? N : nt / 2), m1 = (size_t)r2 * r3;
        P.f_t1.resize((size_t)r1 * m1);
        for (int k1 = 0; k1 < r1; ++k1)
            for (size_t m = 0; m < m1; ++m) {
                const double a = -2.0 * pi * (double)((m * (size_t)k1) % Nc) / (double)Nc;
                P.f_t1[(size_t)k1 * m1 + m] = c32{(float)std::cos(a), (float)std::sin(a)};
            }
        P.f_t2.resize((size_t)r2 * 8);
        for (int k2 = 0; k2 < r2; ++k2)
            for (int j3 = 0; j3 < 8; ++j3) {
                const double a = -2.0 * pi * (double)((j3 * k2) % (r2 * 8)) / (double)(r2 * 8);
                P.f_t2[(size_t)k2 * 8 + j3] = c32{(float)std::cos(a), (float)std::sin(a)};
            }
        P.f_w2n.resize(Nc);
        for (size_t k = 0; k < Nc; ++k) {
            const double a = -pi * (double)k / (double)Nc;
            P.f_w2n[k] = c32{(float)std::cos(a), (float)std::sin(a)};
        }
    }
    P.p_t1.clear(); P.p_t2.clear();
    int q1, q2, q3;
    if (allow_f && allow_p && p_factors(nt, q1, q2, q3)) {
        // the chirp-z tables above stay (stage entry points without a P kernel fall back to them)
        P.family = kFamilyP;
        static const char *const kVariants[] = {"p-mixed-radix-7x11x13-regs-lds", "p-mixed-radix-10x10x10-regs-lds",
                                                "p-mixed-radix-10x10x12-regs-lds", "p-mixed-radix-10x10x15-regs-lds",
                                                "p-mixed-radix-10x10x20-regs-lds"};
        P.variant = kVariants[nt == 1001 ? 0 : nt == 1000 ? 1 : nt == 1200 ? 2 : nt == 1500 ? 3 : 4];
        const size_t m1 = (size_t)q2 * q3;
        P.p_t1.resize(nt);
        for (int k1 = 0; k1 < q1; ++k1)
            for (size_t m = 0; m < m1; ++m) {
                const double a = -2.0 * pi * (double)((m * (size_t)k1) % nt) / (double)nt;
                P.p_t1[(size_t)k1 * m1 + m] = c32{(float)std::cos(a), (float)std::sin(a)};
            }
        P.p_t2.resize(m1);
        for (int k2 = 0; k2 < q2; ++k2)
            for (int j3 = 0; j3 < q3; ++j3) {
                const double a = -2.0 * pi * (double)((j3 * k2) % (int)m1) / (double)m1;
                P.p_t2[(size_t)k2 * q3 + j3] = c32{(float)std::cos(a), (float)std::sin(a)};
            }
    }
    // PH kernels (fft_ph.hpp): an even length whose half is a P plan runs as a half-length complex transform + split
    // (2000 itself stays on its pair-of-traces P kernels, which carry the complex multiplier and the in-launch sums)
    P.half_n = 0;
    if (allow_f && allow_p && nt % 2 == 0 && nt != 2000 && P.family != kFamilyP && p_factors(nt / 2, q1, q2, q3)) {
        const size_t Nh = nt / 2, m1 = (size_t)q2 * q3;
        P.half_n = (int)Nh;
        static const char *const kHalf[] = {"ph-half-length-mixed-radix-7x11x13-regs-lds", "ph-half-length-mixed-radix-10x10x10-regs-lds",
                                            "ph-half-length-mixed-radix-10x10x12-regs-lds", "ph-half-length-mixed-radix-10x10x15-regs-lds",
                                            "ph-half-length-mixed-radix-10x10x20-regs-lds"};
        P.variant = kHalf[Nh == 1001 ? 0 : Nh == 1000 ? 1 : Nh == 1200 ? 2 : Nh == 1500 ? 3 : 4];
        P.p_t1.resize(Nh);
        for (int k1 = 0; k1 < q1; ++k1)
            for (size_t m = 0; m < m1; ++m) {
                const double a = -2.0 * pi * (double)((m * (size_t)k1) % Nh) / (double)Nh;
                P.p_t1[(size_t)k1 * m1 + m] = c32{(float)std::cos(a), (float)std::sin(a)};
            }
        const size_t t2e = m1 + (m1 & 1);
        P.p_t2.assign(t2e + Nh / 2 + 1, c32{0.0f, 0.0f});
        for (int k2 = 0; k2 < q2; ++k2)
            for (int j3 = 0; j3 < q3; ++j3) {
                const double a = -2.0 * pi * (double)((j3 * k2) % (int)m1) / (double)m1;
                P.p_t2[(size_t)k2 * q3 + j3] = c32{(float)std::cos(a), (float)std::sin(a)};
            }
        for (size_t k = 0; k <= Nh / 2; ++k) {
            const double a = -pi * (double)k / (double)Nh;
            P.p_t2[t2e + k] = c32{(float)std::cos(a), (float)std::sin(a)};
        }
    }
    return true;
}

inline PlanDev plan_dev(const PlanHost &H, const c32 *tw, const c32 *tw_split,
                        const c32 *chirp_conj, const c32 *bfft, const c32 *f_t1 = nullptr,
                        const c32 *f_t2 = nullptr, const c32 *f_w2n = nullptr,
                        const float *ones = nullptr, const c32 *p_t1 = nullptr, const c32 *p_t2 = nullptr,
                        c32 *big_scratch = nullptr)
{
    PlanDev D;
    D.nt = H.nt;
    D.nf = H.nf;
    D.mode = H.mode;
    D.log2n = H.log2n;
    D.buf_entries = H.buf_entries;
    D.lds_per_wave = H.lds_per_wave;
    D.waves_per_block = H.waves_per_block;
    D.tw = tw;
    D.tw_split = tw_split;
    D.chirp_conj = chirp_conj;
    D.bfft = bfft;
    D.family = kFamilyG;
    if (H.family == kFamilyF && f_t1 && f_t2 && f_w2n && ones) D.family = kFamilyF;
    if ((H.family == kFamilyFB || H.family == kFamilyFB2 || H.family == kFamilyFB4 || H.family == kFamilyFB8) && f_t1 && f_t2 && ones
        && chirp_conj && bfft)
        D.family = H.family;
    if (H.family == kFamilyP && p_t1 && p_t2 && ones) D.family = kFamilyP;
    D.p_t1 = p_t1;
    D.p_t2 = p_t2;
    D.half_n = (H.half_n && p_t1 && p_t2 && ones) ? H.half_n : 0;
    D.big_scratch = H.big ? big_scratch : nullptr;
    D.big_waves = H.big_waves;
    D.ones = ones;
    D.f_t1 = f_t1;
    D.f_t2 = f_t2;
    D.f_w2n = f_w2n;
    return D;
}

}  // namespace thz
