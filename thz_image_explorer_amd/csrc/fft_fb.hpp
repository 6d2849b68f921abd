// fft_fb.hpp — "FB" kernels: the fused default chain for trace lengths that are not a power of
// two (real instrument scans: nt ~ 1000), as a chirp-z (Bluestein) transform over the
// register-resident complex FFT core of the F family (fft_f.hpp).
//
//   X[k] = w[k] * sum_n (x[n] w[n]) conj(w)[k-n],   w[n] = exp(-i pi n^2 / nt)
//
// i.e. one circular convolution of length M >= 2 nt - 1 (M = 512 / 1024 / 2048 = the complex
// sizes of the three F plans) = two M-point FFTs through f_core_pass1/f_core_pass23; the inverse
// real transform is the same machinery applied to conj(Y) (y = Re DFT(conj Y) / nt).  Two real
// traces share each convolution (packed as x1 + i x2), so a trace costs two M-point FFTs like a
// power-of-two trace of 2 M samples — for a quarter of its bytes: the kernel is VALU-bound, not
// HBM-bound (algorithmic bytes are 16 nt + 20 per trace as for the power-of-two chain).
//
// Same parity rules as the G kernels it replaces for these lengths (kernels.hip k_fft_fwd /
// k_fft_inv): DC (and Nyquist for even nt) imaginary parts are zero, phases are taken before the
// band-pass, the masked spectrum feeds the inverse.
#pragma once

namespace thz {

struct FBArgs {
    size_t npix;
    int nt, nf;
    const float *in;        // (npix, nt)            (forward / fused)
    const cx *fft_in;       // (npix, nf)            (inverse only)
    const float *pre_win;   // nt or nullptr
    const float *mask;      // nf (never null: the plan's ones vector stands in)
    const float *post_win;  // nt or nullptr
    cx *fft_out;            // (npix, nf)
    float *amp_out, *ph_out;
    float *data_out;        // (npix, nt)
    float *img;             // npix or nullptr
    const cx *w;            // nt: exp(-i pi n^2 / nt)
    const cx *bf;           // M : FFT_M(b) / M
    const cx *cmask;        // nf complex multipliers on top of `mask`, or nullptr (P kernels)
    float *sum_partial;     // (gridDim.x, 2 nf): every block's sums of its traces' stored amplitudes | unwrapped phases (k_p<..., SUMS>)
};

// LDS per block, in floats behind the core tables: [mask nf][pre nt][post nt], each padded to 4
template <class P>
struct FBLayout {
    static constexpr int pad4(int v) { return (v + 3) & ~3; }
    static size_t lds_bytes(int waves, int nt, int nf)
    {
        return (size_t)(P::T1_ENTRIES + P::T2_ENTRIES + waves * P::WAVE_ENTRIES) * sizeof(cx)
               + (size_t)(pad4(nf) + 2 * pad4(nt)) * sizeof(float);
    }
};

// base[idx] with the byte offset formed in 32 bits: the load then takes the uniform base in SGPRs
// and one offset VGPR (global_load ... v_off, s[base]) instead of a 64-bit address pair per element
template <class T>
__device__ __forceinline__ T ld_off(const T *base, unsigned idx)
{
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + (size_t)(idx * (unsigned)sizeof(T)));
}

// r[c][j1] <- swap(A[n] * bf[n]), n = M1 j1 + C1 lane + c, A in buf in the nat() layout
template <class P>
__device__ __forceinline__ void fb_multiply_swapped(const cx *buf, const cx *__restrict__ bf, int lane,
                                                    cx (&r)[P::C1][P::R1])
{
    constexpr int R1 = P::R1, C1 = P::C1, M1 = P::M1;
    int fbase[2][C1];
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int c = 0; c < C1; ++c) fbase[v][c] = launder_v(nat(M1 * v + C1 * lane + c) - M1 * v);
    const unsigned bl = (unsigned)launder_v(C1 * lane);
#pragma unroll
    for (int j1 = 0; j1 < R1; ++j1) {
#pragma unroll
        for (int c = 0; c < C1; ++c) {
            const cx a = buf[fbase[j1 & 1][c] + M1 * j1];
            const cx t = cx_mul(a, ld_off(bf, bl + (unsigned)(M1 * j1 + c)));
            r[c][j1] = cx{t.y, t.x};
        }
        if ((j1 & 3) == 3) THZ_SCHED_FENCE();
    }
}

// One spectrum's share of the epilogue for the four bins of a lane: |X| m, arg X, numpy_unwrap
// (two-level scan over ascending bins, state carried across the groups), stores, and the masked
// value for the inverse.
struct FBUnwrap {
    float carry = 0.0f, prev_tail = 0.0f, first = 0.0f;
    float a[4] = {0.0f, 0.0f, 0.0f, 0.0f}, y[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // the group just finished: stored amplitudes and
                                                                            // unwrapped phases of this lane's four bins (0 where
                                                                            // the bin does not exist) — for the in-launch pixel sums
};

// f / ao / po may each be null (stage entry points that do not want that output)
// unwrapped phases of four consecutive bins per lane (bin 256 g + 4 lane + c), running over the groups of a trace
__device__ __forceinline__ void fb_phase_bins(const cx (&X)[4], const bool (&ok)[4], int g, int lane, FBUnwrap &u,
                                              float *po)
{
    const float kPi = 3.14159274101257324219f, kTwoPi = 2.0f * kPi;
    float ph[4];
    {
        // four angles as two packed Horner chains (same values as four fast_atan2f calls, half the issue slots)
        const float yy[4] = {X[0].y, X[1].y, X[2].y, X[3].y}, xx[4] = {X[0].x, X[1].x, X[2].x, X[3].x};
        fast_atan2f_x4(yy, xx, ph);
    }
    if (g == 0) u.first = wave_bcast<0>(ph[0]);
    float prev = wave_shr1(ph[3]);
    if (lane == 0) prev = u.prev_tail;
    float s_[4], run = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float d = ph[c] - (c == 0 ? prev : ph[c - 1]);
        d += (d > kPi) ? -kTwoPi : ((d < -kPi) ? kTwoPi : 0.0f);  // numpy_unwrap's if / else if
        if ((g == 0 && c == 0 && lane == 0) || !ok[c]) d = 0.0f;
        run += d;
        s_[c] = run;
    }
    const float incl = wave_scan_add(run);
    const float excl = wave_shr1(incl);
    const float base = u.carry + excl;
    u.carry += wave_bcast<kWave - 1>(incl);
    u.prev_tail = wave_bcast<kWave - 1>(ph[3]);
#pragma unroll
    for (int c = 0; c < 4; ++c) u.y[c] = ok[c] ? u.first + (base + s_[c]) : 0.0f;
    if (!po) return;
    if (ok[3]) {
        store_f4(po, u.first + (base + s_[0]), u.first + (base + s_[1]), u.first + (base + s_[2]),
                 u.first + (base + s_[3]));
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (ok[c]) po[c] = u.first + (base + s_[c]);
    }
}

// stores of a group of bins: spectrum Y (f), amplitudes a (ao)
__device__ __forceinline__ void fb_store_bins(const cx (&Y)[4], const float (&a)[4], const bool (&ok)[4], cx *f, float *ao)
{
    if (ok[3]) {
        if (f) {
            store_f4(reinterpret_cast<float *>(f), Y[0].x, Y[0].y, Y[1].x, Y[1].y);
            store_f4(reinterpret_cast<float *>(f) + 4, Y[2].x, Y[2].y, Y[3].x, Y[3].y);
        }
        if (ao) store_f4(ao, a[0], a[1], a[2], a[3]);
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (ok[c]) {
                if (f) f[c] = Y[c];
                if (ao) ao[c] = a[c];
            }
    }
}

// real multiplier m (the band pass): spectrum X m, amplitude |X| m, phase of X
__device__ __forceinline__ void fb_finish_bins(const cx (&X)[4], const float (&m)[4], const bool (&ok)[4], int g,
                                               int lane, FBUnwrap &u, cx *f, float *ao, float *po)
{
    float a[4];
    cx Y[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        a[c] = fast_sqrt(fmaf(X[c].x, X[c].x, X[c].y * X[c].y)) * m[c];
        Y[c] = cx{X[c].x * m[c], X[c].y * m[c]};
    }
    fb_store_bins(Y, a, ok, f, ao);
#pragma unroll
    for (int c = 0; c < 4; ++c) u.a[c] = ok[c] ? a[c] : 0.0f;
    fb_phase_bins(X, ok, g, lane, u, po);
}

// complex multiplier h (K13: band pass x reference deconvolution): spectrum Y = X h with the imaginary part
// dropped where `real_bin` (DC / Nyquist: the C2R precondition, math_tools.rs:510-512), amplitude |X h| (taken
// before the dropping), phase of X — as the F kernels' CMASK epilogue.  Y is handed back for the inverse.
__device__ __forceinline__ void fb_finish_bins_c(const cx (&X)[4], const cx (&h)[4], const bool (&real_bin)[4],
                                                 const bool (&ok)[4], int g, int lane, FBUnwrap &u, cx *f, float *ao,
                                                 float *po, cx (&Y)[4])
{
    float a[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        Y[c] = cx_mul(X[c], h[c]);
        a[c] = fast_sqrt(fmaf(Y[c].x, Y[c].x, Y[c].y * Y[c].y));
        if (real_bin[c]) Y[c].y = 0.0f;
    }
    fb_store_bins(Y, a, ok, f, ao);
#pragma unroll
    for (int c = 0; c < 4; ++c) u.a[c] = ok[c] ? a[c] : 0.0f;
    fb_phase_bins(X, ok, g, lane, u, po);
}

// Two real traces per convolution: z = (x1 + i x2) w goes through the chirp-z machinery once,
// F[k] = X1[k] + i X2[k] is split with F[nt-k] (w[nt-k] = (-1)^nt w[k], so no second table read),
// and the inverse transforms conj(Y1full + i Y2full): y1 = Re U / nt, y2 = -Im U / nt.  Four
// M-point FFTs per pair instead of per trace.
template <class P, int MODE>
__global__ __launch_bounds__(512) void k_fb(FBArgs A, FTables T)
{
    THZ_DYN_LDS(lds);
    constexpr int R1 = P::R1, C1 = P::C1, M1 = P::M1;
    const int L = A.nt, nf = A.nf;
    const int lane = lane_id();
    const int wib = THZ_UNIFORM((int)(threadIdx.x >> 6));
    const int wpb = (int)(blockDim.x >> 6);
    cx *t1 = reinterpret_cast<cx *>(lds);
    cx *t2 = t1 + P::T1_ENTRIES;
    cx *buf = t2 + P::T2_ENTRIES + (size_t)wib * P::WAVE_ENTRIES;
    float *mask_s = reinterpret_cast<float *>(t2 + P::T2_ENTRIES + (size_t)wpb * P::WAVE_ENTRIES);
    float *pre_s = mask_s + FBLayout<P>::pad4(nf);
    float *post_s = pre_s + FBLayout<P>::pad4(L);
    for (int i = (int)threadIdx.x; i < P::T1_ENTRIES; i += (int)blockDim.x) t1[i] = T.t1[i];
    for (int i = (int)threadIdx.x; i < P::T2_ENTRIES; i += (int)blockDim.x) t2[i] = T.t2[i];
    for (int i = (int)threadIdx.x; i < nf; i += (int)blockDim.x) mask_s[i] = A.mask[i];
    for (int i = (int)threadIdx.x; i < L; i += (int)blockDim.x) {
        pre_s[i] = A.pre_win ? A.pre_win[i] : 1.0f;
        post_s[i] = A.post_win ? A.post_win[i] : 1.0f;
    }
    __syncthreads();

    FAddr<P> ad;
    ad.init(lane);
    const DivConst by_nt((float)L);
    const int n_groups = (nf + 255) / 256;  // epilogue groups of 256 bins: bin = 256 g + 4 lane + c
    const int half = L / 2;
    const float sgn = (L & 1) ? -1.0f : 1.0f;  // w[nt-k] = sgn * w[k]
    const int y2_base = (L + 4) & ~3;          // Y2[k] lives at nat(y2_base + k): behind everything F uses
    const size_t n_pairs = (A.npix + 1) / 2;
    const size_t stride = (size_t)gridDim.x * wpb;

    for (size_t q = (size_t)blockIdx.x * wpb + wib; q < n_pairs; q += stride) {
        const size_t p = 2 * q;
        const bool has2 = p + 1 < A.npix;  // wave-uniform
        cx r[C1][R1];
        ad.refresh();
        const cx *wl = launder_uniform(A.w);
        const cx *bf = launder_uniform(A.bf);
        const float *pre_l = launder_uniform((const float *)pre_s);
        const float *post_l = launder_uniform((const float *)post_s);
        const float *mask_l = launder_uniform((const float *)mask_s);
        // lane parts of the indices, opaque per trace: otherwise every clamped index, compare mask and
        // address of the unrolled loops below is hoisted out of the trace loop and lives (spills) forever
        const int lb = launder_v(C1 * lane), lb4 = launder_v(4 * lane), lb1 = launder_v(lane);

        if constexpr (MODE != kInv) {
        // ---- a[n] = (x1[n] + i x2[n]) pre[n] w[n] (zero from nt on), in the core's input layout.
        // Branch-free: indices are clamped and the value selected, so that the loads of a trace are
        // issued together instead of one exec-masked round trip each
        {
            const float *x1 = A.in + p * (size_t)L;
            const float *x2 = has2 ? x1 + L : x1;
            constexpr int H = R1 / 2;  // two batches: 4 registers per element in flight, not for all 32
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float xa[C1][H], xb[C1][H];
                cx wv[C1][H];
#pragma unroll
                for (int j = 0; j < H; ++j) {
#pragma unroll
                    for (int c = 0; c < C1; ++c) {
                        const int n = M1 * (H * h + j) + lb + c;
                        const unsigned nn = (unsigned)(n < L ? n : L - 1);
                        xa[c][j] = ld_off(x1, nn);
                        xb[c][j] = ld_off(x2, nn);
                        wv[c][j] = ld_off(wl, nn);
                    }
                }
#pragma unroll
                for (int j = 0; j < H; ++j) {
#pragma unroll
                    for (int c = 0; c < C1; ++c) {
                        const int n = M1 * (H * h + j) + lb + c;
                        const int nn = n < L ? n : L - 1;
                        const float pw = n < L ? pre_l[nn] : 0.0f;
                        const cx z = cx{xa[c][j] * pw, has2 ? xb[c][j] * pw : 0.0f};
                        r[c][H * h + j] = cx_mul(z, wv[c][j]);
                    }
                }
                THZ_SCHED_FENCE();
            }
        }
        f_core_pass1<P>(r, buf, t1, ad, lane);
        f_core_pass23<P>(buf, t2, ad, lane);
        fb_multiply_swapped<P>(buf, bf, lane, r);
        wave_sync();  // every lane has read A before the core overwrites buf
        f_core_pass1<P>(r, buf, t1, ad, lane);
        f_core_pass23<P>(buf, t2, ad, lane);  // buf = swap(c), nat layout

        // ---- spectrum epilogue: F[k] = w[k] c[k]; X1 = (F[k] + conj F[nt-k]) / 2, X2 = (F[k] - conj F[nt-k]) / 2i
        {
            FBUnwrap u1, u2;
            for (int g = 0; g < n_groups; ++g) {
                const int k0 = 256 * g + lb4;
                cx X1[4], X2[4];
                float m[4];
                bool ok[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int k = k0 + c;
                    ok[c] = k < nf;
                    const int kc = ok[c] ? k : nf - 1;
                    const int km = kc == 0 ? 0 : L - kc;  // F[nt] = F[0]
                    const cx wk = ld_off(wl, (unsigned)kc);
                    const cx s = buf[nat(kc)], sm = buf[nat(km)];
                    const cx Fk = cx_mul(cx{s.y, s.x}, wk);
                    const cx wm = kc == 0 ? wk : cx{sgn * wk.x, sgn * wk.y};
                    const cx Fm = cx_mul(cx{sm.y, sm.x}, wm);
                    // conj(Fm) = (Fm.x, -Fm.y)
                    X1[c] = cx{0.5f * (Fk.x + Fm.x), 0.5f * (Fk.y - Fm.y)};
                    // (Fk - conj Fm) / 2i = (-i/2) (dx + i dy) = (dy/2, -dx/2)
                    X2[c] = cx{0.5f * (Fk.y + Fm.y), -0.5f * (Fk.x - Fm.x)};
                    m[c] = mask_l[kc];
                    if (kc == 0 || ((L & 1) == 0 && kc == nf - 1)) {  // real input: DC / Nyquist bins are real
                        X1[c].y = 0.0f;
                        X2[c].y = 0.0f;
                    }
                }
                const size_t o1 = p * (size_t)nf + k0;
                fb_finish_bins(X1, m, ok, g, lane, u1, A.fft_out ? A.fft_out + o1 : nullptr,
                               A.amp_out ? A.amp_out + o1 : nullptr, A.ph_out ? A.ph_out + o1 : nullptr);
                if (has2)
                    fb_finish_bins(X2, m, ok, g, lane, u2, A.fft_out ? A.fft_out + o1 + nf : nullptr,
                                   A.amp_out ? A.amp_out + o1 + nf : nullptr, A.ph_out ? A.ph_out + o1 + nf : nullptr);
                // masked spectra for the inverse: Y1[k] over c[k] — only its owner reads slot k or
                // nt-k — and Y2[k] behind everything F uses
                if constexpr (MODE == kPipe) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (ok[c]) {
                            buf[nat(k0 + c)] = cx{X1[c].x * m[c], X1[c].y * m[c]};
                            // a missing second trace is exactly zero, as in the stand-alone inverse
                            buf[nat(y2_base + k0 + c)] = has2 ? cx{X2[c].x * m[c], X2[c].y * m[c]} : cx{0.0f, 0.0f};
                        }
                }
            }
        }
        wave_sync();
        } else {
            // inverse only: the two spectra from memory into the slots the fused chain leaves them in;
            // DC (and Nyquist for even nt) imaginary parts are ignored like realfft's C2R does
            const cx *f1 = A.fft_in + p * (size_t)nf;
            for (int k = lb1; k < nf; k += kWave) {
                cx y1 = ld_off(f1, (unsigned)k);
                cx y2 = has2 ? ld_off(f1, (unsigned)(nf + k)) : cx{0.0f, 0.0f};
                if (k == 0 || ((L & 1) == 0 && k == nf - 1)) {
                    y1.y = 0.0f;
                    y2.y = 0.0f;
                }
                buf[nat(k)] = y1;
                buf[nat(y2_base + k)] = y2;
            }
            wave_sync();
        }
        if constexpr (MODE == kFwd) continue;

        // ---- inverse: a'[n] = conj(Y1full[n] + i Y2full[n]) w[n];  Yfull[n] = Y[n] (n <= nt/2), conj(Y[nt-n]) above
        {
            constexpr int H = R1 / 2;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                cx wv[C1][H];
#pragma unroll
                for (int j = 0; j < H; ++j)
#pragma unroll
                    for (int c = 0; c < C1; ++c) {
                        const int n = M1 * (H * h + j) + lb + c;
                        wv[c][j] = ld_off(wl, (unsigned)(n < L ? n : L - 1));
                    }
#pragma unroll
                for (int j = 0; j < H; ++j) {
#pragma unroll
                    for (int c = 0; c < C1; ++c) {
                        const int n = M1 * (H * h + j) + lb + c;
                        const int nn = n < L ? n : L - 1;
                        const bool low = nn <= half;
                        const int kk = low ? nn : L - nn;
                        cx y1 = buf[nat(kk)], y2 = buf[nat(y2_base + kk)];
                        // Yfull = low ? Y : conj(Y);  G = Y1full + i Y2full;  conj(G) = conj(Y1full) - i conj(Y2full)
                        // low : conj(Y1) - i conj(Y2) = (y1.x - y2.y, -y1.y - y2.x)
                        // high: Y1 - i Y2             = (y1.x + y2.y,  y1.y - y2.x)
                        const cx gc = low ? cx{y1.x - y2.y, -y1.y - y2.x} : cx{y1.x + y2.y, y1.y - y2.x};
                        const cx v = cx_mul(gc, wv[c][j]);
                        r[c][H * h + j] = n < L ? v : cx{0.0f, 0.0f};
                    }
                }
                THZ_SCHED_FENCE();
            }
        }
        wave_sync();
        f_core_pass1<P>(r, buf, t1, ad, lane);
        f_core_pass23<P>(buf, t2, ad, lane);
        fb_multiply_swapped<P>(buf, bf, lane, r);
        wave_sync();
        f_core_pass1<P>(r, buf, t1, ad, lane);
        f_core_pass23<P>(buf, t2, ad, lane);

        // ---- U[n] = w[n] c'[n]:  y1 = Re U / nt, y2 = -Im U / nt, each times post[n]; images = sum y^2
        {
            float *o1 = A.data_out + p * (size_t)L;
            float acc1 = 0.0f, acc2 = 0.0f;
#pragma unroll 4
            for (int n = lb1; n < L; n += kWave) {
                const cx s = buf[nat(n)];
                const cx wv = ld_off(wl, (unsigned)n);
                const cx U = cx_mul(cx{s.y, s.x}, wv);
                const float pw = post_l[n];
                const float v1 = by_nt(U.x) * pw;
                o1[n] = v1;
                acc1 += v1 * v1;
                if (has2) {
                    const float v2 = by_nt(-U.y) * pw;
                    o1[L + n] = v2;
                    acc2 += v2 * v2;
                }
            }
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

// ---------------------------------------------------------------------------------------------
// Lengths 1024 < nt < 8192 that are not a power of two: the convolution length M = S * 2048 is
// S = 2, 4 or 8 times the largest complex size of the F core, so each M-point transform is S core
// runs plus one radix-S stage, arranged so that no reordering pass is needed:
//   forward  : decimation in frequency.  First stage, straight from the loads:
//                b_s[k] = (sum_{q < S/2} f[k + N q] W_S^(q s)) W_M^(s k)
//              (f[m] = 0 for m >= nt and nt <= N S / 2: for S = 2 simply b_0 = f, b_1 = f W_M^k);
//              S core runs leave A[S j + s] in region s
//   multiply : by FFT_M(b)[S j + s] / M, in place, swapped (inverse through the forward passes)
//   inverse  : decimation in time: S core runs on the regions as they are; the last stage
//                c[k + N q] = swap(sum_s W_S^(s q) W_M^(s k) D_s[k])
//              is evaluated where c is consumed (only indices < nt are ever needed).
// Two traces per convolution as in k_fb.
struct FB2Args {
    FBArgs a;       // same fields as the single-core kernel (w: nt, bf: M)
    const cx *tw;   // W_M^m, m < M
};

// r[c][j1] <- swap(region[nat(n)] * bf[S n + s]), n in the core's input layout
template <class P, int S>
__device__ __forceinline__ void fbs_multiply_swapped(const cx *region, const cx *__restrict__ bf, int s, int lane,
                                                     cx (&r)[P::C1][P::R1])
{
    constexpr int R1 = P::R1, C1 = P::C1, M1 = P::M1;
    int fbase[2][C1];
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int c = 0; c < C1; ++c) fbase[v][c] = launder_v(nat(M1 * v + C1 * lane + c) - M1 * v);
    const unsigned bl = (unsigned)launder_v(S * C1 * lane + s);
#pragma unroll
    for (int j1 = 0; j1 < R1; ++j1) {
#pragma unroll
        for (int c = 0; c < C1; ++c) {
            const cx a = region[fbase[j1 & 1][c] + M1 * j1];
            const cx t = cx_mul(a, ld_off(bf, bl + (unsigned)(S * (M1 * j1 + c))));
            r[c][j1] = cx{t.y, t.x};
        }
        if ((j1 & 3) == 3) THZ_SCHED_FENCE();
    }
}

// c[m], m = k + N q, q < S/2
template <class P, int S>
__device__ __forceinline__ cx fbs_c(const cx *reg0, const cx *tw, int m)
{
    constexpr int N = P::N, RS = P::WAVE_ENTRIES;
    const int k = m & (N - 1), q = m >> 11;
    const int slot = nat(k);
    cx z = reg0[slot];
#pragma unroll
    for (int s = 1; s < S; ++s) {
        cx t = cx_mul(ld_off(tw, (unsigned)(s * k)), reg0[s * RS + slot]);
        // W_S^(s q) = W_M^(N * ((s q) mod S)): a wave-uniform table entry
        if (q) t = cx_mul(t, ld_off(tw, (unsigned)(N * ((s * q) & (S - 1)))));
        z = z + t;
    }
    return cx{z.y, z.x};
}

// ---------------------------------------------------------------------------------------------
// FBC: S waves per pair of traces.  If one wave owned all S regions of its pair, LDS (8 regions per
// CU) would cap a CU at 8 / S waves (measured: 2.7 / 14.5 / 25 ms against 2.5 / 7.3 / 5.7 ms at
// nt = 2000 / 4000 / 5000 for the three S).  The S sub-transforms of a stage are independent, so
// wave s of a pair runs sub-transform s in
// region s, and the block (8 waves = 8 / S pairs) meets at a barrier where the stages meet — after
// the inverse core runs, before the on-demand last stage reads all S regions.  The epilogue is
// split by bin groups (two groups per wave); the unwrap's running sum crosses waves as per-group
// totals published in LDS and re-added in group order, so every value is the one the single-wave
// kernels produce.  Forward and inverse are separate kernels (the inverse reads the masked spectra
// from memory), mask read from memory.
template <class P, int S>
struct FBCLayout {
    static constexpr int kWaves = 8, kPairs = 8 / S;
    static constexpr int kXch = 2 * kPairs * (2 * 16 + 2 * S);  // floats: group totals [pair][spectrum][16], image partials
    static size_t lds_bytes()
    {
        return (size_t)(P::T1_ENTRIES + P::T2_ENTRIES + kWaves * P::WAVE_ENTRIES) * sizeof(cx) + (size_t)kXch * sizeof(float);
    }
};

template <class P, int S, int MODE>
__global__ __launch_bounds__(512) void k_fbc(FB2Args B, FTables T)
{
    static_assert(MODE == kFwd || MODE == kInv, "forward and inverse are separate launches");
    THZ_DYN_LDS(lds);
    constexpr int N = P::N, R1 = P::R1, C1 = P::C1, M1 = P::M1, RS = P::WAVE_ENTRIES;
    static_assert(N == 2048, "regions of 2048");
    constexpr int PAIRS = FBCLayout<P, S>::kPairs;
    const FBArgs &A = B.a;
    const int L = A.nt, nf = A.nf;
    const int lane = lane_id();
    const int wib = THZ_UNIFORM((int)(threadIdx.x >> 6));
    const int pi = wib / S, s = wib % S;  // pair slot in the block, sub-transform of this wave
    cx *t1 = reinterpret_cast<cx *>(lds);
    cx *t2 = t1 + P::T1_ENTRIES;
    cx *reg0 = t2 + P::T2_ENTRIES + (size_t)pi * S * RS;  // the pair's regions
    cx *my_reg = reg0 + s * RS;
    float *xch = reinterpret_cast<float *>(t2 + P::T2_ENTRIES + (size_t)8 * RS);
    float *tot = xch + pi * 32;                               // [spectrum][group] totals of the pair
    float *imgp = xch + PAIRS * 32 + pi * 2 * S;              // [spectrum][wave] image partial sums
    for (int i = (int)threadIdx.x; i < P::T1_ENTRIES; i += (int)blockDim.x) t1[i] = T.t1[i];
    for (int i = (int)threadIdx.x; i < P::T2_ENTRIES; i += (int)blockDim.x) t2[i] = T.t2[i];
    __syncthreads();

    FAddr<P> ad;
    ad.init(lane);
    const DivConst by_nt((float)L);
    const int half = L / 2;
    const float sgn = (L & 1) ? -1.0f : 1.0f;  // w[nt-k] = sgn * w[k]
    const float kPi = 3.14159274101257324219f, kTwoPi = 2.0f * kPi;
    const size_t n_pairs = (A.npix + 1) / 2;
    const size_t per_round = (size_t)gridDim.x * PAIRS;
    const size_t rounds = (n_pairs + per_round - 1) / per_round;  // block-uniform trip count: barriers inside

    for (size_t it = 0; it < rounds; ++it) {
        const size_t q = (it * gridDim.x + blockIdx.x) * PAIRS + pi;
        const bool valid = q < n_pairs;  // wave-uniform
        const size_t p = 2 * q;
        const bool has2 = valid && p + 1 < A.npix;
        ad.refresh();
        const cx *wl = launder_uniform(A.w);
        const cx *bf = launder_uniform(A.bf);
        const cx *tw = launder_uniform(B.tw);
        const float *mask_g = launder_uniform(A.mask);
        const float *pre_g = A.pre_win ? launder_uniform(A.pre_win) : nullptr;
        const float *post_g = A.post_win ? launder_uniform(A.post_win) : nullptr;
        const int lb = launder_v(C1 * lane), lb4 = launder_v(4 * lane), lb1 = launder_v(lane);

        auto f_at = [&](int m) -> cx {
            const int mm = m < L ? m : L - 1;
            cx v;
            if constexpr (MODE == kFwd) {
                const float *x1 = A.in + p * (size_t)L;
                const float pw = pre_g ? ld_off(pre_g, (unsigned)mm) : 1.0f;
                const float xa = ld_off(x1, (unsigned)mm);
                const float xb = has2 ? ld_off(x1, (unsigned)(L + mm)) : 0.0f;
                v = cx_mul(cx{xa * pw, xb * pw}, ld_off(wl, (unsigned)mm));
            } else {
                const cx *f1 = A.fft_in + p * (size_t)nf;
                const bool low = mm <= half;
                const int kk = low ? mm : L - mm;
                cx y1 = ld_off(f1, (unsigned)kk);
                cx y2 = has2 ? ld_off(f1, (unsigned)(nf + kk)) : cx{0.0f, 0.0f};
                if (kk == 0 || ((L & 1) == 0 && kk == nf - 1)) {  // realfft's C2R ignores these
                    y1.y = 0.0f;
                    y2.y = 0.0f;
                }
                const cx gc = low ? cx{y1.x - y2.y, -y1.y - y2.x} : cx{y1.x + y2.y, y1.y - y2.x};
                v = cx_mul(gc, ld_off(wl, (unsigned)mm));
            }
            return m < L ? v : cx{0.0f, 0.0f};
        };

        if (valid) {
            cx r[C1][R1];
            // ---- first stage of sub-transform s, its core run, the multiply, the inverse core run:
            // all inside region s, no other wave involved
            constexpr int H = R1 / 4;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int lbh = h ? launder_after(lb, r[C1 - 1][H * h - 1].x) : lb;  // batch after batch
#pragma unroll
                for (int j = 0; j < H; ++j) {
#pragma unroll
                    for (int c = 0; c < C1; ++c) {
                        const int k = M1 * (H * h + j) + lbh + c;
                        cx v = f_at(k);
#pragma unroll
                        for (int qq = 1; qq < S / 2; ++qq) {
                            cx t = f_at(k + N * qq);
                            if (s) t = cx_mul(t, ld_off(tw, (unsigned)(N * ((qq * s) & (S - 1)))));
                            v = v + t;
                        }
                        if (s) v = cx_mul(v, ld_off(tw, (unsigned)(s * k)));
                        r[c][H * h + j] = v;
                    }
                }
                THZ_SCHED_FENCE();
            }
            f_core_pass1<P>(r, my_reg, t1, ad, lane);
            f_core_pass23<P>(my_reg, t2, ad, lane);
            fbs_multiply_swapped<P, S>(my_reg, bf, s, lane, r);
            wave_sync();
            f_core_pass1<P>(r, my_reg, t1, ad, lane);
            f_core_pass23<P>(my_reg, t2, ad, lane);
        }
        __syncthreads();  // D_0 .. D_{S-1} of every pair are complete

        if constexpr (MODE == kFwd) {
            // ---- spectrum epilogue, two groups of 256 bins per wave: groups 2 s and 2 s + 1
            const int n_groups = (nf + 255) / 256;
            cx X1[2][4], X2[2][4];
            float mk[2][4], sr1[2][4], sr2[2][4], ex1[2], ex2[2];
            float first1 = 0.0f, first2 = 0.0f;
            auto spectra_at = [&](int kc, cx &x1v, cx &x2v) {
                const int km = kc == 0 ? 0 : L - kc;  // F[nt] = F[0]
                const cx wk = ld_off(wl, (unsigned)kc);
                const cx Fk = cx_mul(fbs_c<P, S>(reg0, tw, kc), wk);
                const cx wm = kc == 0 ? wk : cx{sgn * wk.x, sgn * wk.y};
                const cx Fm = cx_mul(fbs_c<P, S>(reg0, tw, km), wm);
                x1v = cx{0.5f * (Fk.x + Fm.x), 0.5f * (Fk.y - Fm.y)};
                x2v = cx{0.5f * (Fk.y + Fm.y), -0.5f * (Fk.x - Fm.x)};
                if (kc == 0 || ((L & 1) == 0 && kc == nf - 1)) {
                    x1v.y = 0.0f;
                    x2v.y = 0.0f;
                }
            };
            if (valid) {
                {   // raw phase of bin 0 (every wave needs it; cheaper to evaluate than to pass around)
                    cx a0, b0;
                    spectra_at(0, a0, b0);
                    first1 = fast_atan2f(a0.y, a0.x);
                    first2 = fast_atan2f(b0.y, b0.x);
                }
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    const int g = 2 * s + gg;
                    if (g < n_groups) {
                        const int k0 = 256 * g + lb4;
                        float ph1[4], ph2[4];
                        bool ok[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int k = k0 + c;
                            ok[c] = k < nf;
                            const int kc = ok[c] ? k : nf - 1;
                            spectra_at(kc, X1[gg][c], X2[gg][c]);
                            mk[gg][c] = ld_off(mask_g, (unsigned)kc);
                            ph1[c] = fast_atan2f(X1[gg][c].y, X1[gg][c].x);
                            ph2[c] = fast_atan2f(X2[gg][c].y, X2[gg][c].x);
                        }
                        // raw phase of the bin before this group (lane 0 only uses it): the last bin of
                        // the previous group, evaluated here so that the differences need no other wave
                        float pt1 = 0.0f, pt2 = 0.0f;
                        if (g > 0) {
                            cx a0, b0;
                            spectra_at(256 * g - 1, a0, b0);
                            pt1 = fast_atan2f(a0.y, a0.x);
                            pt2 = fast_atan2f(b0.y, b0.x);
                        }
                        float prev1 = wave_shr1(ph1[3]), prev2 = wave_shr1(ph2[3]);
                        if (lane == 0) { prev1 = pt1; prev2 = pt2; }
                        float run1 = 0.0f, run2 = 0.0f;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            float d1 = ph1[c] - (c == 0 ? prev1 : ph1[c - 1]);
                            float d2 = ph2[c] - (c == 0 ? prev2 : ph2[c - 1]);
                            d1 += (d1 > kPi) ? -kTwoPi : ((d1 < -kPi) ? kTwoPi : 0.0f);
                            d2 += (d2 > kPi) ? -kTwoPi : ((d2 < -kPi) ? kTwoPi : 0.0f);
                            if ((g == 0 && c == 0 && lane == 0) || !ok[c]) { d1 = 0.0f; d2 = 0.0f; }
                            run1 += d1; run2 += d2;
                            sr1[gg][c] = run1; sr2[gg][c] = run2;
                        }
                        const float incl1 = wave_scan_add(run1), incl2 = wave_scan_add(run2);
                        ex1[gg] = wave_shr1(incl1);
                        ex2[gg] = wave_shr1(incl2);
                        if (lane == kWave - 1) {
                            tot[g] = incl1;
                            tot[16 + g] = incl2;
                        }
                    }
                }
            }
            __syncthreads();  // the pair's group totals are published
            if (valid) {
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    const int g = 2 * s + gg;
                    if (g < n_groups) {
                        float carry1 = 0.0f, carry2 = 0.0f;  // the single-wave kernels' carry += total, group by group
                        for (int gp = 0; gp < g; ++gp) {
                            carry1 += tot[gp];
                            carry2 += tot[16 + gp];
                        }
                        const int k0 = 256 * g + lb4;
                        const size_t o1 = p * (size_t)nf + k0;
                        const float base1 = carry1 + ex1[gg], base2 = carry2 + ex2[gg];
                        auto put = [&](const cx(&X)[4], const float(&sr)[4], float first, float base, size_t o) {
                            const float(&m)[4] = mk[gg];
                            float am[4], py[4];
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                am[c] = fast_sqrt(fmaf(X[c].x, X[c].x, X[c].y * X[c].y)) * m[c];
                                py[c] = first + (base + sr[c]);
                            }
                            if (k0 + 3 < nf) {
                                if (A.fft_out) {
                                    float *f = reinterpret_cast<float *>(A.fft_out + o);
                                    store_f4(f, X[0].x * m[0], X[0].y * m[0], X[1].x * m[1], X[1].y * m[1]);
                                    store_f4(f + 4, X[2].x * m[2], X[2].y * m[2], X[3].x * m[3], X[3].y * m[3]);
                                }
                                if (A.amp_out) store_f4(A.amp_out + o, am[0], am[1], am[2], am[3]);
                                if (A.ph_out) store_f4(A.ph_out + o, py[0], py[1], py[2], py[3]);
                            } else {
#pragma unroll
                                for (int c = 0; c < 4; ++c)
                                    if (k0 + c < nf) {
                                        if (A.fft_out) A.fft_out[o + c] = cx{X[c].x * m[c], X[c].y * m[c]};
                                        if (A.amp_out) A.amp_out[o + c] = am[c];
                                        if (A.ph_out) A.ph_out[o + c] = py[c];
                                    }
                            }
                        };
                        put(X1[gg], sr1[gg], first1, base1, o1);
                        if (has2) put(X2[gg], sr2[gg], first2, base2, o1 + nf);
                    }
                }
            }
        } else {
            // ---- U[n] = w[n] c'[n] over this wave's share of the samples; image partial sums meet in LDS
            float acc1 = 0.0f, acc2 = 0.0f;
            if (valid) {
                const int chunk = (((L + S - 1) / S + kWave - 1) / kWave) * kWave;
                const int n_end = (s + 1) * chunk < L ? (s + 1) * chunk : L;
                float *o1 = A.data_out + p * (size_t)L;
#pragma unroll 2
                for (int n = s * chunk + lb1; n < n_end; n += kWave) {
                    const cx U = cx_mul(fbs_c<P, S>(reg0, tw, n), ld_off(wl, (unsigned)n));
                    const float pw = post_g ? ld_off(post_g, (unsigned)n) : 1.0f;
                    const float v1 = by_nt(U.x) * pw;
                    o1[n] = v1;
                    acc1 += v1 * v1;
                    if (has2) {
                        const float v2 = by_nt(-U.y) * pw;
                        o1[L + n] = v2;
                        acc2 += v2 * v2;
                    }
                }
                acc1 = wave_reduce_add(acc1);
                acc2 = wave_reduce_add(acc2);
                if (lane == 0) {
                    imgp[s] = acc1;
                    imgp[S + s] = acc2;
                }
            }
            __syncthreads();
            if (valid && A.img && s == 0 && lane == 0) {
                float a1 = 0.0f, a2 = 0.0f;
                for (int w = 0; w < S; ++w) {
                    a1 += imgp[w];
                    a2 += imgp[S + w];
                }
                A.img[p] = a1;
                if (has2) A.img[p + 1] = a2;
            }
        }
        __syncthreads();  // regions and exchange slots are reused by the next round
    }
}

}  // namespace thz
