// kernels.hip — gfx950 kernels of the THz per-pixel engine and their launchers.
//
// "G" family (this file): one trace per wavefront, the whole transform staged
// in that wave's private slice of LDS, Stockham radix-4/2 autosort passes, no
// workgroup barriers (waves never wait for each other).  It handles every
// supported trace length: powers of two directly (real transform as a
// half-length complex transform + split), anything else through Bluestein's
// chirp-z over the same wave-level complex FFT.
//
// Reference arithmetic being replaced (paths relative to the reference):
//   window + R2C + |.| + arg + unwrap   src/math_tools.rs:330-398, 211-240
//   frequency band-pass multiply        src/filters/band_pass_fd.rs:184-212
//   C2R + 1/nt                          src/math_tools.rs:545-568
//   time band-pass multiply             src/filters/band_pass_td_*.rs:155-175
//   intensity image                     src/data_thread.rs:1288-1307
//   pixel means                         src/math_tools.rs:421-440
//   ROI mask / mean                     src/math_tools.rs:574-661
//   block-mean scaling                  src/math_tools.rs:273-301
//   load-time bias subtraction          src/io.rs:578-596
#include "kernels.hpp"
#include "fft_f.hpp"
#include "fft_fb.hpp"
#include "fft_p.hpp"
#include "fft_ph.hpp"

#include <cstdlib>

namespace thz {

// Where a wave's two transform buffers live decides what orders its accesses: in LDS the DS instructions of one wave
// execute in issue order and wave_sync() (a compiler fence) is all it takes; in GLOBAL scratch — trace lengths whose
// buffers do not fit the CU's LDS: not a power of two above 8191, powers of two above 16384 (round 3) — a lane's loads
// must see what other lanes of the wave stored in the pass before: a workgroup-scope fence (the waves of a block share the
// CU's vector cache, whatever else the compiler knows the target needs).
struct SyncLds {
    static __device__ __forceinline__ void sync() { wave_sync(); }
};
struct SyncGlobal {
    static __device__ __forceinline__ void sync()
    {
#ifdef THZ_EMU
        wave_sync();
#else
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#endif
    }
};

// ---------------------------------------------------------------------------
// wave-level complex FFT in LDS (Stockham autosort, radix-4 then radix-2)
// a, b: ping-pong buffers of N = 1 << log2n entries, input in a.
// tw:   W_N^m = exp(-2*pi*i*m/N), m in [0, N)
// Returns the buffer that holds the result.  Ends with a wave_sync().
// ---------------------------------------------------------------------------
template <class S = SyncLds>
__device__ __forceinline__ c32 *wave_cfft(c32 *a, c32 *b, int log2n, const c32 *__restrict__ tw,
                                          int lane)
{
    const int N = 1 << log2n;
    c32 *src = a, *dst = b;
    int log2ns = 0;
    int rem = log2n;
    while (rem >= 2) {
        const int Ns = 1 << log2ns;
        const int q4 = N >> 2;
        const int tws = N >> (log2ns + 2);  // N / (4*Ns)
        for (int t = lane; t < q4; t += kWave) {
            const int k = t & (Ns - 1);
            c32 u0 = src[t], u1 = src[t + q4], u2 = src[t + 2 * q4], u3 = src[t + 3 * q4];
            if (log2ns > 0) {
                u1 = cmul(u1, tw[k * tws]);
                u2 = cmul(u2, tw[2 * k * tws]);
                u3 = cmul(u3, tw[3 * k * tws]);
            }
            const c32 s02 = cadd(u0, u2), d02 = csub(u0, u2);
            const c32 s13 = cadd(u1, u3), d13 = csub(u1, u3);
            const int j = ((t - k) << 2) + k;
            dst[j] = cadd(s02, s13);
            dst[j + Ns] = c32{d02.re + d13.im, d02.im - d13.re};
            dst[j + 2 * Ns] = csub(s02, s13);
            dst[j + 3 * Ns] = c32{d02.re - d13.im, d02.im + d13.re};
        }
        S::sync();
        c32 *t_ = src; src = dst; dst = t_;
        log2ns += 2;
        rem -= 2;
    }
    if (rem == 1) {
        const int Ns = 1 << log2ns;
        const int q2 = N >> 1;
        const int tws = N >> (log2ns + 1);  // N / (2*Ns)
        for (int t = lane; t < q2; t += kWave) {
            const int k = t & (Ns - 1);
            c32 u0 = src[t], u1 = src[t + q2];
            if (log2ns > 0) u1 = cmul(u1, tw[k * tws]);
            const int j = ((t - k) << 1) + k;
            dst[j] = cadd(u0, u1);
            dst[j + Ns] = csub(u0, u1);
        }
        S::sync();
        c32 *t_ = src; src = dst; dst = t_;
    }
    return src;
}

// Bluestein core: on entry `a` holds the chirp-premultiplied, zero-padded
// sequence (M entries).  Leaves c = IFFT_M(FFT_M(a) .* bfft) SWAPPED (re<->im)
// in the returned buffer; bfft already carries 1/M.
template <class S = SyncLds>
__device__ __forceinline__ c32 *wave_bluestein_core(c32 *a, c32 *b, const PlanDev &P, int lane)
{
    const int M = 1 << P.log2n;
    c32 *Z = wave_cfft<S>(a, b, P.log2n, P.tw, lane);
    c32 *other = (Z == a) ? b : a;
    for (int i = lane; i < M; i += kWave) {
        const c32 v = cmul(Z[i], P.bfft[i]);
        Z[i] = c32{v.im, v.re};  // swap: inverse transform through the forward passes
    }
    S::sync();
    return wave_cfft<S>(Z, other, P.log2n, P.tw, lane);
}

// ---------------------------------------------------------------------------
// Shared epilogue of the forward transform: Xb holds the nf spectrum bins in
// natural order; scratch is an LDS area of >= nf floats no longer in use.
// Writes fft (masked), |X| (masked), unwrapped phase (never masked).
// When KEEP_MASKED the masked spectrum is also written back into Xb (the fused
// pipeline feeds it to the inverse transform).
// ---------------------------------------------------------------------------
template <bool KEEP_MASKED, class S = SyncLds>
__device__ __forceinline__ void spectrum_epilogue(c32 *Xb, float *scratch, int nf, size_t p,
                                                  c32 *__restrict__ fft_out,
                                                  float *__restrict__ amp_out,
                                                  float *__restrict__ ph_out,
                                                  const float *__restrict__ mask, int lane)
{
    const bool want_phase = ph_out != nullptr;
    for (int k = lane; k < nf; k += kWave) {
        const c32 X = Xb[k];
        const float m = mask ? mask[k] : 1.0f;
        if (amp_out) {
            const float a = sqrtf(fmaf(X.re, X.re, X.im * X.im));
            amp_out[p * nf + k] = mask ? a * m : a;
        }
        if (want_phase) scratch[k] = atan2f(X.im, X.re);
        const c32 Xm = mask ? c32{X.re * m, X.im * m} : X;
        if (fft_out) fft_out[p * nf + k] = Xm;
        if (KEEP_MASKED) Xb[k] = Xm;
    }
    S::sync();
    if (!want_phase) return;

    // numpy_unwrap (math_tools.rs:211-240) as a two-level scan: each lane owns
    // a contiguous chunk, corrections are decided on raw successive
    // differences exactly as the recurrence does.
    const float kPi = 3.14159274101257324219f;
    const float kTwoPi = 2.0f * kPi;
    const int C = (nf + kWave - 1) / kWave;
    const int start = lane * C;
    const int end = (start + C < nf) ? start + C : nf;
    float prev = (start > 0 && start < nf) ? scratch[start - 1] : 0.0f;
    const float first = scratch[0];
    S::sync();
    float s = 0.0f;
    for (int i = start; i < end; ++i) {
        const float v = scratch[i];
        float d = 0.0f;
        if (i > 0) {
            d = v - prev;
            if (d > kPi) d -= kTwoPi;
            else if (d < -kPi) d += kTwoPi;
        }
        prev = v;
        s += d;
        scratch[i] = s;
    }
    const float incl = wave_scan_add(s);
    const float excl = wave_shr1(incl);
    for (int i = start; i < end; ++i) scratch[i] = first + (excl + scratch[i]);
    S::sync();
    for (int k = lane; k < nf; k += kWave) ph_out[p * nf + k] = scratch[k];
    S::sync();
}

// R2C split for the power-of-two path: Z = FFT_N(z), z[n] = x[2n] + i x[2n+1]
// -> X[0..N] into Xb (N+1 entries).
template <class S = SyncLds>
__device__ __forceinline__ void r2c_split(const c32 *Z, c32 *Xb, int N,
                                          const c32 *__restrict__ tw_split, int lane)
{
    for (int k = lane; k <= N / 2; k += kWave) {
        if (k == 0) {
            const c32 z = Z[0];
            Xb[0] = c32{z.re + z.im, 0.0f};
            Xb[N] = c32{z.re - z.im, 0.0f};
        } else {
            const c32 a = Z[k], b = Z[N - k];
            const c32 E = c32{0.5f * (a.re + b.re), 0.5f * (a.im - b.im)};
            const c32 O = c32{0.5f * (a.im + b.im), -0.5f * (a.re - b.re)};
            const c32 t = cmul(O, tw_split[k]);
            Xb[k] = cadd(E, t);
            Xb[N - k] = cconj(csub(E, t));
        }
    }
    S::sync();
}

// Inverse of r2c_split for the unnormalised C2R, written SWAPPED (re<->im) so
// that the forward passes compute the inverse transform.
template <class S = SyncLds>
__device__ __forceinline__ void c2r_merge_swapped(const c32 *X, c32 *Zs, int N,
                                                  const c32 *__restrict__ tw_split, int lane)
{
    for (int k = lane; k <= N / 2; k += kWave) {
        if (k == 0) {
            const float x0 = X[0].re, xn = X[N].re;
            Zs[0] = c32{x0 - xn, x0 + xn};  // swapped {im, re}
        } else {
            const c32 a = X[k], b = X[N - k];
            const c32 E = c32{a.re + b.re, a.im - b.im};
            const c32 D = c32{a.re - b.re, a.im + b.im};
            const c32 w = tw_split[k];
            const c32 O = c32{D.re * w.re + D.im * w.im, D.im * w.re - D.re * w.im};
            // Z[k] = E + iO ; Z[N-k] = conj(E - iO)
            const c32 zk = c32{E.re - O.im, E.im + O.re};
            const c32 zn = c32{E.re + O.im, -(E.im - O.re)};
            Zs[k] = c32{zk.im, zk.re};
            Zs[N - k] = c32{zn.im, zn.re};
        }
    }
    S::sync();
}

__device__ __forceinline__ float apply2(float v, const float *__restrict__ wa,
                                        const float *__restrict__ wb, int i)
{
    if (wa) v *= wa[i];
    if (wb) v *= wb[i];
    return v;
}

// Final stage of the inverse for the power-of-two path: R holds the swapped
// result; x[2n] = R[n].im, x[2n+1] = R[n].re; /nt, * window, store, sum of
// squares.
__device__ __forceinline__ void c2r_store(const c32 *R, int N, int nt, size_t p,
                                          const float *__restrict__ win, float *__restrict__ out,
                                          float *__restrict__ img, int lane)
{
    const DivConst by_nt((float)nt);
    float acc = 0.0f;
    for (int n = lane; n < N; n += kWave) {
        const c32 r = R[n];
        float v0 = by_nt(r.im), v1 = by_nt(r.re);
        if (win) { v0 *= win[2 * n]; v1 *= win[2 * n + 1]; }
        *reinterpret_cast<float2 *>(out + p * nt + 2 * n) = make_float2(v0, v1);
        acc += v0 * v0;
        acc += v1 * v1;
    }
    if (img) {
        acc = wave_reduce_add(acc);
        if (lane == 0) img[p] = acc;
    }
}

// ---------------------------------------------------------------------------
// forward:  math_tools::fft (+ optional band-pass multiply)
// ---------------------------------------------------------------------------
template <class S>
__device__ __forceinline__ void fft_fwd_body(const PlanDev &P, size_t npix, const float *__restrict__ in,
                                             const float *__restrict__ wa, const float *__restrict__ wb,
                                             float *__restrict__ data_out, c32 *__restrict__ fft_out,
                                             float *__restrict__ amp_out, float *__restrict__ ph_out,
                                             const float *__restrict__ mask, c32 *A, c32 *B)
{
    const int lane = lane_id();
    const int wib = (int)(threadIdx.x >> 6);
    const int wpb = (int)(blockDim.x >> 6);
    const int N = 1 << P.log2n;
    const int nt = P.nt, nf = P.nf;

    for (size_t p = (size_t)blockIdx.x * wpb + wib; p < npix; p += (size_t)gridDim.x * wpb) {
        const float *x = in + p * nt;
        c32 *Xb;
        float *scratch;
        if (P.mode == kModePow2) {
            for (int i = lane; i < N; i += kWave) {
                float2 v = *reinterpret_cast<const float2 *>(x + 2 * i);
                v.x = apply2(v.x, wa, wb, 2 * i);
                v.y = apply2(v.y, wa, wb, 2 * i + 1);
                if (data_out) *reinterpret_cast<float2 *>(data_out + p * nt + 2 * i) = v;
                A[i] = c32{v.x, v.y};
            }
            S::sync();
            c32 *Z = wave_cfft<S>(A, B, P.log2n, P.tw, lane);
            Xb = (Z == A) ? B : A;
            r2c_split<S>(Z, Xb, N, P.tw_split, lane);
            scratch = reinterpret_cast<float *>(Z);
        } else {
            for (int i = lane; i < N; i += kWave) {
                c32 v = c32{0.0f, 0.0f};
                if (i < nt) {
                    const float xv = apply2(x[i], wa, wb, i);
                    if (data_out) data_out[p * nt + i] = xv;
                    const c32 c = P.chirp_conj[i];
                    v = c32{xv * c.re, xv * c.im};
                }
                A[i] = v;
            }
            S::sync();
            c32 *R = wave_bluestein_core<S>(A, B, P, lane);
            Xb = (R == A) ? B : A;
            for (int k = lane; k < nf; k += kWave) {
                const c32 r = R[k];
                Xb[k] = cmul(c32{r.im, r.re}, P.chirp_conj[k]);
            }
            S::sync();
            if (lane == 0) {
                Xb[0].im = 0.0f;  // real input: DC (and Nyquist) bins are real
                if ((nt & 1) == 0) Xb[nf - 1].im = 0.0f;
            }
            S::sync();
            scratch = reinterpret_cast<float *>(R);
        }
        spectrum_epilogue<false, S>(Xb, scratch, nf, p, fft_out, amp_out, ph_out, mask, lane);
    }
}

__global__ __launch_bounds__(256) void k_fft_fwd(PlanDev P, size_t npix,
                                                 const float *__restrict__ in,
                                                 const float *__restrict__ wa,
                                                 const float *__restrict__ wb,
                                                 float *__restrict__ data_out,
                                                 c32 *__restrict__ fft_out,
                                                 float *__restrict__ amp_out,
                                                 float *__restrict__ ph_out,
                                                 const float *__restrict__ mask)
{
    THZ_DYN_LDS(lds);
    c32 *A = reinterpret_cast<c32 *>(lds + (size_t)(threadIdx.x >> 6) * P.lds_per_wave);
    fft_fwd_body<SyncLds>(P, npix, in, wa, wb, data_out, fft_out, amp_out, ph_out, mask, A, A + P.buf_entries);
}

// the same transform with the wave's buffers in global scratch (P.big_scratch: 2 buf_entries per wave of the grid)
__global__ __launch_bounds__(256) void k_fft_fwd_big(PlanDev P, size_t npix,
                                                     const float *__restrict__ in,
                                                     const float *__restrict__ wa,
                                                     const float *__restrict__ wb,
                                                     float *__restrict__ data_out,
                                                     c32 *__restrict__ fft_out,
                                                     float *__restrict__ amp_out,
                                                     float *__restrict__ ph_out,
                                                     const float *__restrict__ mask)
{
    const size_t gw = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    c32 *A = P.big_scratch + gw * 2 * (size_t)P.buf_entries;
    fft_fwd_body<SyncGlobal>(P, npix, in, wa, wb, data_out, fft_out, amp_out, ph_out, mask, A, A + P.buf_entries);
}

// ---------------------------------------------------------------------------
// inverse:  math_tools::ifft per-pixel part (+ optional window, intensity)
// ---------------------------------------------------------------------------
template <class S = SyncLds>
__device__ __forceinline__ void inverse_from_lds(const PlanDev &P, c32 *X, c32 *other, size_t p,
                                                 const float *__restrict__ win,
                                                 float *__restrict__ out, float *__restrict__ img,
                                                 int lane)
{
    const int N = 1 << P.log2n;
    // pow2 only: X holds nf = N+1 bins in natural order
    c2r_merge_swapped<S>(X, other, N, P.tw_split, lane);
    c32 *R = wave_cfft<S>(other, X, P.log2n, P.tw, lane);
    c2r_store(R, N, P.nt, p, win, out, img, lane);
    S::sync();
}

template <class S>
__device__ __forceinline__ void fft_inv_body(const PlanDev &P, size_t npix, const c32 *__restrict__ fft_in,
                                             const float *__restrict__ win, float *__restrict__ out,
                                             float *__restrict__ img, c32 *A, c32 *B)
{
    const int lane = lane_id();
    const int wib = (int)(threadIdx.x >> 6);
    const int wpb = (int)(blockDim.x >> 6);
    const int N = 1 << P.log2n;
    const int nt = P.nt, nf = P.nf;

    for (size_t p = (size_t)blockIdx.x * wpb + wib; p < npix; p += (size_t)gridDim.x * wpb) {
        const c32 *Xg = fft_in + p * nf;
        if (P.mode == kModePow2) {
            for (int k = lane; k < nf; k += kWave) A[k] = Xg[k];
            S::sync();
            inverse_from_lds<S>(P, A, B, p, win, out, img, lane);
        } else {
            // x[t] = Re DFT(conj Xfull)[t]; Xfull = Hermitian extension
            const int half = nt / 2;
            for (int j = lane; j < N; j += kWave) {
                c32 v = c32{0.0f, 0.0f};
                if (j < nt) {
                    const int kk = (j <= half) ? j : nt - j;
                    c32 X = Xg[kk];
                    if (kk == 0 || ((nt & 1) == 0 && kk == half)) X.im = 0.0f;
                    // conj(Xfull[j]): for j <= half conj(X), else conj(conj X) = X
                    if (j <= half) X.im = -X.im;
                    v = cmul(X, P.chirp_conj[j]);
                }
                A[j] = v;
            }
            S::sync();
            c32 *R = wave_bluestein_core<S>(A, B, P, lane);
            const DivConst by_nt((float)nt);
            float acc = 0.0f;
            for (int t = lane; t < nt; t += kWave) {
                const c32 r = R[t];
                const c32 c = P.chirp_conj[t];
                // Re( swap(r) * c ) = r.im*c.re - r.re*c.im
                float v = by_nt(r.im * c.re - r.re * c.im);
                if (win) v *= win[t];
                out[p * nt + t] = v;
                acc += v * v;
            }
            if (img) {
                acc = wave_reduce_add(acc);
                if (lane == 0) img[p] = acc;
            }
            S::sync();
        }
    }
}

__global__ __launch_bounds__(256) void k_fft_inv(PlanDev P, size_t npix,
                                                 const c32 *__restrict__ fft_in,
                                                 const float *__restrict__ win,
                                                 float *__restrict__ out, float *__restrict__ img)
{
    THZ_DYN_LDS(lds);
    c32 *A = reinterpret_cast<c32 *>(lds + (size_t)(threadIdx.x >> 6) * P.lds_per_wave);
    fft_inv_body<SyncLds>(P, npix, fft_in, win, out, img, A, A + P.buf_entries);
}

__global__ __launch_bounds__(256) void k_fft_inv_big(PlanDev P, size_t npix,
                                                     const c32 *__restrict__ fft_in,
                                                     const float *__restrict__ win,
                                                     float *__restrict__ out, float *__restrict__ img)
{
    const size_t gw = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    c32 *A = P.big_scratch + gw * 2 * (size_t)P.buf_entries;
    fft_inv_body<SyncGlobal>(P, npix, fft_in, win, out, img, A, A + P.buf_entries);
}

// ---------------------------------------------------------------------------
// fused default chain (power-of-two nt): one launch, one HBM pass
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pipeline(PlanDev P, size_t npix,
                                                  const float *__restrict__ raw,
                                                  const float *__restrict__ pre_win,
                                                  const float *__restrict__ mask,
                                                  const float *__restrict__ post_win,
                                                  c32 *__restrict__ fft_out,
                                                  float *__restrict__ amp_out,
                                                  float *__restrict__ ph_out,
                                                  float *__restrict__ data_out,
                                                  float *__restrict__ img)
{
    THZ_DYN_LDS(lds);
    const int lane = lane_id();
    const int wib = (int)(threadIdx.x >> 6);
    const int wpb = (int)(blockDim.x >> 6);
    const int N = 1 << P.log2n;
    const int nt = P.nt, nf = P.nf;
    c32 *A = reinterpret_cast<c32 *>(lds + (size_t)wib * P.lds_per_wave);
    c32 *B = A + P.buf_entries;

    for (size_t p = (size_t)blockIdx.x * wpb + wib; p < npix; p += (size_t)gridDim.x * wpb) {
        const float *x = raw + p * nt;
        for (int i = lane; i < N; i += kWave) {
            float2 v = *reinterpret_cast<const float2 *>(x + 2 * i);
            if (pre_win) { v.x *= pre_win[2 * i]; v.y *= pre_win[2 * i + 1]; }
            A[i] = c32{v.x, v.y};
        }
        wave_sync();
        c32 *Z = wave_cfft(A, B, P.log2n, P.tw, lane);
        c32 *Xb = (Z == A) ? B : A;
        r2c_split(Z, Xb, N, P.tw_split, lane);
        spectrum_epilogue<true>(Xb, reinterpret_cast<float *>(Z), nf, p, fft_out, amp_out, ph_out,
                                mask, lane);
        inverse_from_lds(P, Xb, Z, p, post_win, data_out, img, lane);
    }
}

// ---------------------------------------------------------------------------
// elementwise / reduction kernels
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fd_mask(size_t npix, int nf, c32 *__restrict__ fft,
                                                 float *__restrict__ amp,
                                                 const float *__restrict__ mask)
{
    const size_t total = npix * (size_t)nf;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const float m = mask[i % nf];
        if (fft) { c32 v = fft[i]; fft[i] = c32{v.re * m, v.im * m}; }
        if (amp) amp[i] = amp[i] * m;
    }
}

__global__ __launch_bounds__(256) void k_fd_cmask(size_t npix, int nf, int nt,
                                                  c32 *__restrict__ fft, float *__restrict__ amp,
                                                  const c32 *__restrict__ cmask)
{
    const size_t total = npix * (size_t)nf;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % nf);
        const c32 m = cmask[k];
        if (fft) {
            c32 v = cmul(fft[i], m);
            if (k == 0 || ((nt & 1) == 0 && k == nf - 1)) v.im = 0.0f;
            fft[i] = v;
        }
        if (amp) amp[i] = amp[i] * sqrtf(fmaf(m.re, m.re, m.im * m.im));
    }
}

// small vectors: out = in * f; dst += src (pixel-mean finalisation, same-device all-reduce of group_api.cpp)
__global__ __launch_bounds__(256) void k_scale_vec(const float *__restrict__ in, float f, size_t n, float *__restrict__ out)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i] * f;
}
__global__ __launch_bounds__(256) void k_add_vec(float *__restrict__ dst, const float *__restrict__ src, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] += src[i];
}
__global__ __launch_bounds__(256) void k_add_u64(unsigned long long *__restrict__ dst, const unsigned long long *__restrict__ src, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

// out = in * win.  One wave per trace, 16-byte accesses when the rows allow it; no per-element
// index division (the earlier flat grid-stride form spent its time in a 64-bit modulo).
template <bool VEC>
__global__ __launch_bounds__(256) void k_td_window(size_t npix, int nt,
                                                   const float *__restrict__ in,
                                                   const float *__restrict__ win,
                                                   float *__restrict__ out)
{
    const int lane = lane_id();
    const int wpb = (int)(blockDim.x >> 6);
    for (size_t p = (size_t)blockIdx.x * wpb + (threadIdx.x >> 6); p < npix; p += (size_t)gridDim.x * wpb) {
        const float *x = in + p * (size_t)nt;
        float *y = out + p * (size_t)nt;
        if constexpr (VEC) {
#pragma unroll 4
            for (int e = 4 * lane; e < nt; e += 4 * kWave) {
                const float4 v = *reinterpret_cast<const float4 *>(x + e);
                const float4 w = *reinterpret_cast<const float4 *>(win + e);
                *reinterpret_cast<float4 *>(y + e) = make_float4(v.x * w.x, v.y * w.y, v.z * w.z, v.w * w.w);
            }
        } else {
            for (int e = lane; e < nt; e += kWave) y[e] = x[e] * win[e];
        }
    }
}

// The same for trace lengths that are CH whole rounds of a wave's 16-byte accesses (nt = 256 CH, CH <= 16): a lane meets
// the same CH window chunks in every trace, so it keeps them in registers — the loop then issues one load and one
// store per chunk instead of two loads and a store, and all CH loads of a trace before its first store.
template <int CH>
__global__ __launch_bounds__(256) void k_td_window_regs(size_t npix, const float *__restrict__ in, const float *__restrict__ win,
                                                        float *__restrict__ out)
{
    constexpr int nt = 256 * CH;
    const int lane = lane_id();
    const int wpb = (int)(blockDim.x >> 6);
    float4 w[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) w[c] = *reinterpret_cast<const float4 *>(win + 4 * lane + 256 * c);
    for (size_t p = (size_t)blockIdx.x * wpb + (threadIdx.x >> 6); p < npix; p += (size_t)gridDim.x * wpb) {
        const float *x = in + p * (size_t)nt + 4 * lane;
        float *y = out + p * (size_t)nt + 4 * lane;
        float4 v[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) v[c] = *reinterpret_cast<const float4 *>(x + 256 * c);
#pragma unroll
        for (int c = 0; c < CH; ++c)
            *reinterpret_cast<float4 *>(y + 256 * c) = make_float4(v[c].x * w[c].x, v[c].y * w[c].y, v[c].z * w[c].z, v[c].w * w[c].w);
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void k_intensity(size_t npix, int nt, float *__restrict__ data,
                                                   float *__restrict__ img, int subtract_bias)
{
    const int lane = lane_id();
    const int wib = (int)(threadIdx.x >> 6);
    const int wpb = (int)(blockDim.x >> 6);
    for (size_t p = (size_t)blockIdx.x * wpb + wib; p < npix; p += (size_t)gridDim.x * wpb) {
        float *x = data + p * (size_t)nt;
        const float off = subtract_bias ? x[0] : 0.0f;
        wave_sync();  // every lane has read x[0] before lane 0 overwrites it
        float acc = 0.0f;
        if constexpr (VEC) {
            for (int e = 4 * lane; e < nt; e += 4 * kWave) {
                float4 v = *reinterpret_cast<const float4 *>(x + e);
                if (subtract_bias) {
                    v = make_float4(v.x - off, v.y - off, v.z - off, v.w - off);
                    *reinterpret_cast<float4 *>(x + e) = v;
                }
                acc += v.x * v.x;
                acc += v.y * v.y;
                acc += v.z * v.z;
                acc += v.w * v.w;
            }
        } else {
            for (int i = lane; i < nt; i += kWave) {
                float v = x[i];
                if (subtract_bias) { v = v - off; x[i] = v; }
                acc += v * v;
            }
        }
        if (img) {
            acc = wave_reduce_add(acc);
            if (lane == 0) img[p] = acc;
        }
    }
}

// pixel sums in the reference's order (math_tools.rs:421-440):
// pass 1: acc[y][i] = (sum over x, sequential) / nx   (divide skipped if nx_div == 0)
// carry (or null): the sum continues from it — the rows of the slabs in front, in a group's reference-order means
__global__ __launch_bounds__(256) void k_sum_axis0(const float *__restrict__ arr, size_t n0,
                                                   size_t inner, float div,
                                                   float *__restrict__ out, const float *carry)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < inner;
         i += (size_t)gridDim.x * blockDim.x) {
        // sequential f32 sum over axis 0 (ndarray's order); only the adds are ordered, so the
        // loads of 16 rows go out together
        float s = carry ? carry[i] : 0.0f;
        size_t a = 0;
        const float *col = arr + i;
        for (; a + 16 <= n0; a += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = col[(a + u) * inner];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u];
        }
        for (; a < n0; ++a) s += col[a * inner];
        out[i] = (div > 0.0f) ? s / div : s;
    }
}

// the rows the fused launches leave per block (in-launch pixel sums): few and short, so the adds run in double and
// add nothing to the f32 error of the blocks' own chains
__global__ __launch_bounds__(256) void k_sum_rows_f64(const float *__restrict__ arr, size_t n0, size_t inner,
                                                      float *__restrict__ out)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < inner; i += (size_t)gridDim.x * blockDim.x) {
        double s = 0.0;
        size_t a = 0;
        const float *col = arr + i;
        for (; a + 16 <= n0; a += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = col[(a + u) * inner];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += (double)v[u];
        }
        for (; a < n0; ++a) s += (double)col[a * inner];
        out[i] = (float)s;
    }
}

// Column sums with memory-level parallelism (pixel-mean partials of big cubes):
// one block per row group; thread t owns the 16-byte column chunks t, t+256, ...
// of every row (a full row is read by consecutive chunk-iterations, 4 KiB each)
// and keeps U rows x KC chunks of loads in flight.  Rows are only 4-byte aligned
// (L = 2*nf or nf); the ragged last chunk (L % 4 floats) is summed by thread 0.
// `list` (or null): the rows to add are arr's rows list[0 .. nrows) — a region of interest's pixels (session ROI
// sums); the sums of a list are order-free like the pixel sums.
template <int KC>
__global__ __launch_bounds__(256) void k_colsum_partial(const float *__restrict__ arr,
                                                        size_t nrows, size_t L,
                                                        size_t rows_per_group,
                                                        float *__restrict__ partial,
                                                        const uint32_t *__restrict__ list)
{
    constexpr int U = 2;
    auto row_of = [&](size_t r) -> size_t { return list ? (size_t)list[r] : r; };
    // rows are dealt round-robin to the blocks (row = rg + G*i): at any moment the
    // chip streams one contiguous band of the array, like a linear copy does
    const size_t rg = blockIdx.x, G = gridDim.x;
    (void)rows_per_group;
    const size_t nfull = L / 4;  // complete 16-byte chunks per row
    const size_t t = threadIdx.x;
    float acc[KC][4];
#pragma unroll
    for (int k = 0; k < KC; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[k][i] = 0.0f;
    size_t r = rg;
    for (; r + (U - 1) * G < nrows; r += U * G) {
        float v[U][KC][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t row = row_of(r + u * G);
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                const size_t ch = t + 256 * (size_t)k;
                if (ch < nfull) load_f4(arr + row * L + 4 * ch, v[u][k][0], v[u][k][1], v[u][k][2], v[u][k][3]);
                else v[u][k][0] = v[u][k][1] = v[u][k][2] = v[u][k][3] = 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < KC; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[k][i] += v[u][k][i];
    }
    for (; r < nrows; r += G)
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const size_t ch = t + 256 * (size_t)k;
            if (ch < nfull) {
                float a, b, c, d;
                load_f4(arr + row_of(r) * L + 4 * ch, a, b, c, d);
                acc[k][0] += a; acc[k][1] += b; acc[k][2] += c; acc[k][3] += d;
            }
        }
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const size_t ch = t + 256 * (size_t)k;
        if (ch < nfull)
#pragma unroll
            for (int i = 0; i < 4; ++i) partial[rg * L + 4 * ch + i] = acc[k][i];
    }
    // ragged tail columns: lanes of wave 0 split the rows, then a wave reduction
    const size_t tail0 = nfull * 4;
    if (tail0 < L && t < kWave) {
        for (size_t c = tail0; c < L; ++c) {
            float s = 0.0f;
            for (size_t rr = rg + G * t; rr < nrows; rr += G * kWave) s += arr[row_of(rr) * L + c];
            s = wave_reduce_add(s);
            if (t == 0) partial[rg * L + c] = s;
        }
    }
}

// ROI mask, bit-exact u64 restatement of math_tools.rs:574-591, 604-652
__global__ __launch_bounds__(256) void k_roi_mask(const uint64_t *__restrict__ poly, int n,
                                                  uint64_t x_min, uint64_t x_max, uint64_t y_min,
                                                  uint64_t y_max, uint64_t x_size, uint64_t y_size,
                                                  uint8_t *__restrict__ mask)
{
    const uint64_t total = x_size * y_size;
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t y = idx / x_size, x = idx % x_size;
        bool inside = false;
        if (n > 0 && x >= x_min && x <= x_max && y >= y_min && y <= y_max) {
            int j = n - 1;
            for (int i = 0; i < n; ++i) {
                const uint64_t xi = poly[2 * i], yi = poly[2 * i + 1];
                const uint64_t xj = poly[2 * j], yj = poly[2 * j + 1];
                if ((yi > y) != (yj > y)) {
                    const uint64_t rhs = (xj - xi) * (y - yi) / (yj - yi) + xi;
                    if (x < rhs) inside = !inside;
                }
                j = i;
            }
        }
        mask[idx] = inside ? 1 : 0;
    }
}

// out[z] = sum over listed pixels (in list order) of arr[pix*len + z]
// Sum over a list of pixels, one thread per sample index z, in the list's order — the
// reference's sequential f32 accumulation (math_tools.rs:640-659), so the result is bit-exact.
// The order is a property of the adds only: the loads of kGatherBatch pixels are issued together
// (kGatherBatch rows of 256 B per wave in flight) and then added one after the other.  One wave
// per block so that the nt (or nf) independent chains spread over as many CUs as possible.
constexpr int kGatherBatch = 64;

__global__ __launch_bounds__(64) void k_gather_sum(const float *__restrict__ arr, size_t len,
                                                   const uint32_t *__restrict__ list,
                                                   uint32_t count, float div,
                                                   float *__restrict__ out)
{
    const size_t z = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (z >= len) return;
    const float *col = arr + z;
    float s = 0.0f;
    uint32_t c = 0;
    for (; c + kGatherBatch <= count; c += kGatherBatch) {
        float v[kGatherBatch];
#pragma unroll
        for (int i = 0; i < kGatherBatch; ++i) v[i] = col[(size_t)list[c + i] * len];
#pragma unroll
        for (int i = 0; i < kGatherBatch; ++i) s += v[i];
    }
    for (; c < count; ++c) s += col[(size_t)list[c] * len];
    out[z] = (div > 0.0f) ? s / div : s;
}

// k_gather_sum over arr * w1 * w2 * w3 (each factor optional, applied in this order with its own rounding): the ROI
// mean of the fft stage's `data` output — the input traces after the Tilt taper, "Time Band Pass" and the fft
// window, three successive f32 multiplies in the reference (tilt_compensation.rs:171-201, band_pass_td_*.rs:155-175,
// math_tools.rs:356-371) — summed in the reference's order (math_tools.rs:477-483 -> :640-659) without the windowed
// cube ever being stored.
__global__ __launch_bounds__(64) void k_gather_sum_w(const float *__restrict__ arr, size_t len,
                                                     const uint32_t *__restrict__ list, uint32_t count, float div,
                                                     const float *__restrict__ w1, const float *__restrict__ w2,
                                                     const float *__restrict__ w3, float *__restrict__ out)
{
    const size_t z = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (z >= len) return;
    const float *col = arr + z;
    const float a = w1 ? w1[z] : 1.0f, b = w2 ? w2[z] : 1.0f, c3 = w3 ? w3[z] : 1.0f;
    auto weigh = [&](float v) {
        if (w1) v = v * a;
        if (w2) v = v * b;
        if (w3) v = v * c3;
        return v;
    };
    float s = 0.0f;
    uint32_t c = 0;
    for (; c + kGatherBatch <= count; c += kGatherBatch) {
        float v[kGatherBatch];
#pragma unroll
        for (int i = 0; i < kGatherBatch; ++i) v[i] = col[(size_t)list[c + i] * len];
#pragma unroll
        for (int i = 0; i < kGatherBatch; ++i) s += weigh(v[i]);
    }
    for (; c < count; ++c) s += weigh(col[(size_t)list[c] * len]);
    out[z] = (div > 0.0f) ? s / div : s;
}

// Block means over a slab edge (math_tools.rs:273-301 adds the s x s inputs of a block row by row, column by column,
// and divides by s^2): `m` of the block's s rows are in `arr`; the sum continues from `carry` (the rows in front, summed by
// the previous slab — or null) and is divided only when div > 0 (the block's last rows).  One thread per (block
// column, sample); the same adds in the same order as k_scale3d, the same exact division.
__global__ __launch_bounds__(256) void k_scale_rows_partial(const float *__restrict__ arr, size_t m, size_t ny, size_t L, size_t s,
                                                           const float *__restrict__ carry, float div, float *__restrict__ out)
{
    const size_t nh = ny / s;
    const DivConst by(div > 0.0f ? div : 1.0f);
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < nh * L; t += (size_t)gridDim.x * blockDim.x) {
        const size_t ay = t / L, z = t % L;
        float sum = carry ? carry[t] : 0.0f;
        for (size_t i = 0; i < m; ++i)
            for (size_t j = 0; j < s; ++j) sum += arr[(i * ny + ay * s + j) * L + z];
        out[t] = div > 0.0f ? by(sum) : sum;
    }
}

// out = in / d (IEEE division, what `result[z] /= pixel_counts[z] as f32` does, math_tools.rs:655-659), or
// (in * w) / d when w is given
__global__ __launch_bounds__(256) void k_div_vec(const float *__restrict__ in, const float *__restrict__ w, float d, size_t n,
                                                 float *__restrict__ out)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (w ? in[i] * w[i] : in[i]) / d;
}

// One wave per pixel: the trace is copied to its insert position on the extended axis, the front
// is filled with its first sample, the rest with zeros (tilt_compensation.rs:171-201).
__global__ __launch_bounds__(256) void k_tilt(size_t npix, int nt_in, int nt_out,
                                              const float *__restrict__ in,
                                              const float *__restrict__ taper,
                                              const int *__restrict__ insert_index,
                                              float *__restrict__ out)
{
    const int lane = lane_id();
    const int wpb = (int)(blockDim.x >> 6);
    for (size_t p = (size_t)blockIdx.x * wpb + (threadIdx.x >> 6); p < npix; p += (size_t)gridDim.x * wpb) {
        const int ins = insert_index[p];
        const int end = (ins + nt_in < nt_out) ? ins + nt_in : nt_out;
        const float *raw = in + p * (size_t)nt_in;
        float *o = out + p * (size_t)nt_out;
        const float first = raw[0];
#pragma unroll 8
        for (int e = lane; e < nt_out; e += kWave) {
            float v = 0.0f;
            if (e < ins) v = first;
            else if (e < end) v = raw[e - ins] * taper[e - ins];
            o[e] = v;
        }
    }
}

// Block mean over s x s pixels (math_tools.rs:273-301): one wave per output pixel, sample axis
// across the lanes, the s*s inputs added in the reference's i-outer / j-inner order.
// VEC: rows are whole 16-byte chunks at 16-byte-aligned addresses — four samples per lane and access.
template <bool VEC>
__global__ __launch_bounds__(256) void k_scale3d(const float *__restrict__ arr, size_t nx,
                                                 size_t ny, size_t L, size_t s,
                                                 float *__restrict__ out)
{
    const size_t nw = nx / s, nh = ny / s;
    const DivConst by_ss((float)(s * s));  // the exact quotient (thz_device.hpp), not a reciprocal multiply
    const int lane = lane_id();
    const int wpb = (int)(blockDim.x >> 6);
    for (size_t q = (size_t)blockIdx.x * wpb + (threadIdx.x >> 6); q < nw * nh; q += (size_t)gridDim.x * wpb) {
        const size_t ax = q / nh, ay = q % nh;
        float *o = out + q * L;
        if constexpr (VEC) {
#pragma unroll 2
            for (size_t z = 4 * (size_t)lane; z < L; z += 4 * kWave) {
                float4 sum = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                for (size_t i = 0; i < s; ++i)
                    for (size_t j = 0; j < s; ++j) {
                        const size_t ox = ax * s + i, oy = ay * s + j;
                        if (ox < nx && oy < ny) {
                            const float4 v = *reinterpret_cast<const float4 *>(arr + (ox * ny + oy) * L + z);
                            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
                        }
                    }
                *reinterpret_cast<float4 *>(o + z) = make_float4(by_ss(sum.x), by_ss(sum.y), by_ss(sum.z), by_ss(sum.w));
            }
        } else {
#pragma unroll 4
            for (size_t z = (size_t)lane; z < L; z += kWave) {
                float sum = 0.0f;
                for (size_t i = 0; i < s; ++i)
                    for (size_t j = 0; j < s; ++j) {
                        const size_t ox = ax * s + i, oy = ay * s + j;
                        if (ox < nx && oy < ny) sum += arr[(ox * ny + oy) * L + z];
                    }
                o[z] = by_ss(sum);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Frequency-dependent Richardson–Lucy deconvolution (K12),
// src/filters/deconvolution.rs:766-1041.
//
//  k_dc_fft      every trace, zero-padded to M = next_pow2(nt + 498): real FFT,
//                spectrum kept (one forward transform serves all bands)
//  k_dc_energy   per trace and band b:  |h_b * x|^2 summed over the "same"
//                slice [249, 249 + nt) of the linear convolution (:574-609, :966)
//  k_rl_*        Richardson–Lucy on every band's energy image (:620-712), all
//                bands batched per iteration
//  k_dc_gain     g_b = sqrt(max(u_b, 0) / d_b)  (:975, :986-993)
//  k_dc_combine  out = sum_b g_b (h_b * x) = IFFT(X . sum_b g_b H_b), sliced
//                (:996-1011; linear in x, so one inverse per trace)
// The reference convolves in Complex<f64> and rounds to f32; here the FIR runs
// through the f32 transform (error ~1e-6 of the trace maximum).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dc_fft(PlanDev P, size_t npix, int nt,
                                                const float *__restrict__ in,
                                                c32 *__restrict__ spec)
{
    THZ_DYN_LDS(lds);
    const int lane = lane_id();
    const int wib = (int)(threadIdx.x >> 6);
    const int wpb = (int)(blockDim.x >> 6);
    const int N = 1 << P.log2n;
    c32 *A = reinterpret_cast<c32 *>(lds + (size_t)wib * P.lds_per_wave);
    c32 *B = A + P.buf_entries;
    for (size_t p = (size_t)blockIdx.x * wpb + wib; p < npix; p += (size_t)gridDim.x * wpb) {
        const float *x = in + p * nt;
        for (int i = lane; i < N; i += kWave) {
            const float a = (2 * i < nt) ? x[2 * i] : 0.0f;
            const float b = (2 * i + 1 < nt) ? x[2 * i + 1] : 0.0f;
            A[i] = c32{a, b};
        }
        wave_sync();
        c32 *Z = wave_cfft(A, B, P.log2n, P.tw, lane);
        c32 *Xb = (Z == A) ? B : A;
        r2c_split(Z, Xb, N, P.tw_split, lane);
        for (int k = lane; k <= N; k += kWave) spec[p * (size_t)(N + 1) + k] = Xb[k];
        wave_sync();
    }
}

// value i of the real sequence held (swapped) in R: x[2n] = R[n].im, x[2n+1] = R[n].re
__device__ __forceinline__ float dc_real_at(const c32 *R, int i)
{
    const c32 r = R[i >> 1];
    return (i & 1) ? r.re : r.im;
}

// LDS per wave: X (N+2) | A (N+2) | B (N+2)
__global__ __launch_bounds__(256) void k_dc_energy(PlanDev P, size_t npix, int nt, int n_bands,
                                                   int shift, const c32 *__restrict__ spec,
                                                   const c32 *__restrict__ H,
                                                   float *__restrict__ energy)
{
    THZ_DYN_LDS(lds);
    const int lane = lane_id();
    const int wib = (int)(threadIdx.x >> 6);
    const int wpb = (int)(blockDim.x >> 6);
    const int N = 1 << P.log2n;
    c32 *X = reinterpret_cast<c32 *>(lds + (size_t)wib * (size_t)(3 * P.buf_entries) * sizeof(c32));
    c32 *A = X + P.buf_entries;
    c32 *B = A + P.buf_entries;
    for (size_t p = (size_t)blockIdx.x * wpb + wib; p < npix; p += (size_t)gridDim.x * wpb) {
        for (int k = lane; k <= N; k += kWave) X[k] = spec[p * (size_t)(N + 1) + k];
        wave_sync();
        for (int b = 0; b < n_bands; ++b) {
            const c32 *Hb = H + (size_t)b * (N + 1);
            for (int k = lane; k <= N; k += kWave) A[k] = cmul(X[k], Hb[k]);
            wave_sync();
            c2r_merge_swapped(A, B, N, P.tw_split, lane);
            const c32 *R = wave_cfft(B, A, P.log2n, P.tw, lane);
            float acc = 0.0f;
            for (int t = lane; t < nt; t += kWave) {
                const float v = dc_real_at(R, t + shift);
                acc += v * v;
            }
            acc = wave_reduce_add(acc);
            if (lane == 0) energy[(size_t)b * npix + p] = acc;
            wave_sync();
        }
    }
}

// The same band energies on the register-resident F core (fft_f.hpp), for M = 1024 / 2048 / 4096:
// a pixel's spectrum stays in registers for all bands; per band the product with H_b goes into the
// wave's LDS buffer in the core's nat() layout, the inverse core runs, and the energy over the
// "same" slice is summed straight out of the core's output — no stores except one float per band.
// The next band's multiplier is fetched while the current band's passes run.
// LDS: [T1][T2][w2n head][wg][per wave: N + 2].
template <class PL>
__global__ __launch_bounds__(512) void k_dc_energy_f(FTables T, size_t npix, int nt, int n_bands, int shift,
                                                     const cx *__restrict__ spec, const cx *__restrict__ H,
                                                     float *__restrict__ energy)
{
    THZ_DYN_LDS(lds);
    constexpr int N = PL::N, R1 = PL::R1, C1 = PL::C1;
    const int nf = N + 1;
    const int lane = lane_id();
    const int wib = (int)(threadIdx.x >> 6), wpb = (int)(blockDim.x >> 6);
    cx *t1 = reinterpret_cast<cx *>(lds);
    cx *t2 = t1 + PL::T1_ENTRIES;
    cx *w2n_s = t2 + PL::T2_ENTRIES;
    cx *wg_s = w2n_s + PL::W2N_HEAD;
    cx *buf = wg_s + PL::WG_ENTRIES + (size_t)wib * PL::WAVE_ENTRIES;
    for (int i = (int)threadIdx.x; i < PL::W2N_HEAD; i += (int)blockDim.x) w2n_s[i] = f_stage_w2n(T.w2n[i]);
    if ((int)threadIdx.x < R1) wg_s[threadIdx.x] = T.w2n[PL::M1 * (int)threadIdx.x];
    for (int i = (int)threadIdx.x; i < PL::T1_ENTRIES; i += (int)blockDim.x) t1[i] = T.t1[i];
    for (int i = (int)threadIdx.x; i < PL::T2_ENTRIES; i += (int)blockDim.x) t2[i] = T.t2[i];
    __syncthreads();
    FAddr<PL> ad;
    ad.init(lane);
    // where the spectrum rows / the core's output sit in buf (see k_f's inverse branch, f_time_epilogue)
    const int s2 = (lane >> 4) & 3;
    const int sb2 = (2 * lane) ^ (s2 & 2);
    const bool swap2 = (s2 & 1) != 0;
    const int sb1a = nat(lane), sb1b = nat(kWave + lane) - kWave;
    for (size_t p = (size_t)blockIdx.x * wpb + wib; p < npix; p += (size_t)gridDim.x * wpb) {
        float xs[R1][2 * C1], hs[R1][2 * C1];
        f_load_spec<PL>(spec + p * nf, lane, xs);
        const float x_nyq = spec[p * nf + N].x;
        f_load_spec<PL>(H, lane, hs);
        float h_nyq = H[N].x;
        for (int b = 0; b < n_bands; ++b) {
            ad.refresh();
#pragma unroll
            for (int j = 0; j < R1; ++j) {
                if constexpr (C1 == 2) {
                    const cx e0 = cx_mul(cx{xs[j][0], xs[j][1]}, cx{hs[j][0], hs[j][1]});
                    const cx e1 = cx_mul(cx{xs[j][2], xs[j][3]}, cx{hs[j][2], hs[j][3]});
                    st2(buf + sb2 + 2 * kWave * j, swap2 ? e1 : e0, swap2 ? e0 : e1);
                } else {
                    buf[((j & 1) ? sb1b : sb1a) + kWave * j] = cx_mul(cx{xs[j][0], xs[j][1]}, cx{hs[j][0], hs[j][1]});
                }
            }
            if (lane == 0) buf[N] = cx{x_nyq * h_nyq, 0.0f};
            if (b + 1 < n_bands) {
                f_load_spec<PL>(H + (size_t)(b + 1) * nf, lane, hs);
                h_nyq = H[(size_t)(b + 1) * nf + N].x;
            }
            wave_sync();
            cx r[C1][R1];
            f_inverse_input<PL, false>(buf, w2n_s, wg_s, nullptr, lane, r);
            wave_sync();
            f_core_pass1<PL>(r, buf, t1, ad, lane);
            f_core_pass23<PL>(buf, t2, ad, lane);
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < R1; ++j) {
                float v[2 * C1];
                if constexpr (C1 == 2) {
                    const cx2 rr = ld2(buf + sb2 + 2 * kWave * j);
                    const cx e0 = swap2 ? rr.b : rr.a, e1 = swap2 ? rr.a : rr.b;
                    v[0] = e0.y; v[1] = e0.x; v[2] = e1.y; v[3] = e1.x;
                } else {
                    const cx rr = buf[((j & 1) ? sb1b : sb1a) + kWave * j];
                    v[0] = rr.y; v[1] = rr.x;
                }
                const int t0 = 2 * C1 * (kWave * j + lane) - shift;  // sample index within the "same" slice
#pragma unroll
                for (int i = 0; i < 2 * C1; ++i)
                    if (t0 + i >= 0 && t0 + i < nt) acc += v[i] * v[i];
            }
            acc = wave_reduce_add(acc);
            if (lane == 0) energy[(size_t)b * npix + p] = acc;
            wave_sync();
        }
    }
}

// Round 3: the same work at three waves per SIMD.  k_dc_energy_f keeps the NEXT band's multiplier in 32 registers
// while a band's passes run (206 VGPRs at N = 1024: two waves per SIMD, eight per CU) and is bound by the latency of
// its five LDS round trips per band, not by arithmetic.  Here the multiplier rows are loaded where they are
// consumed (157 VGPRs, no scratch), blocks are four waves, three blocks share a CU: twelve waves hide the L2 round
// trip of the rows and each other's LDS waits.  The band's energy is summed out of the core's output at the top of the
// next trip, so the stores of one band and the loads of the next are issued back to back.
template <class PL>
__global__ __launch_bounds__(256) THZ_WAVES_PER_SIMD(PL::N <= 512 ? 4 : 3) void k_dc_energy_f3(
    FTables T, size_t npix, int nt, int n_bands, int shift, const cx *__restrict__ spec, const cx *__restrict__ H,
    float *__restrict__ energy)
{
    THZ_DYN_LDS(lds);
    constexpr int N = PL::N, R1 = PL::R1, C1 = PL::C1;
    const int nf = N + 1;
    const int lane = lane_id();
    const int wib = (int)(threadIdx.x >> 6), wpb = (int)(blockDim.x >> 6);
    cx *t1 = reinterpret_cast<cx *>(lds);
    cx *t2 = t1 + PL::T1_ENTRIES;
    cx *w2n_s = t2 + PL::T2_ENTRIES;
    cx *wg_s = w2n_s + PL::W2N_HEAD;
    cx *buf = wg_s + PL::WG_ENTRIES + (size_t)wib * PL::WAVE_ENTRIES;
    for (int i = (int)threadIdx.x; i < PL::W2N_HEAD; i += (int)blockDim.x) w2n_s[i] = f_stage_w2n(T.w2n[i]);
    if ((int)threadIdx.x < R1) wg_s[threadIdx.x] = T.w2n[PL::M1 * (int)threadIdx.x];
    for (int i = (int)threadIdx.x; i < PL::T1_ENTRIES; i += (int)blockDim.x) t1[i] = T.t1[i];
    for (int i = (int)threadIdx.x; i < PL::T2_ENTRIES; i += (int)blockDim.x) t2[i] = T.t2[i];
    __syncthreads();
    FAddr<PL> ad;
    ad.init(lane);
    const int s2 = (lane >> 4) & 3;
    const int sb2 = (2 * lane) ^ (s2 & 2);
    const bool swap2 = (s2 & 1) != 0;
    const int sb1a = nat(lane), sb1b = nat(kWave + lane) - kWave;
    for (size_t p = (size_t)blockIdx.x * wpb + wib; p < npix; p += (size_t)gridDim.x * wpb) {
        float xs[R1][2 * C1];
        f_load_spec<PL>(spec + p * nf, lane, xs);
        const float x_nyq = spec[p * nf + N].x;
#pragma unroll 1
        for (int b = 0; b <= n_bands; ++b) {
            if (b > 0) {  // band b - 1: energy over the "same" slice, out of the core's output
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < R1; ++j) {
                    float v[2 * C1];
                    if constexpr (C1 == 2) {
                        const cx2 rr = ld2(buf + sb2 + 2 * kWave * j);
                        const cx e0 = swap2 ? rr.b : rr.a, e1 = swap2 ? rr.a : rr.b;
                        v[0] = e0.y; v[1] = e0.x; v[2] = e1.y; v[3] = e1.x;
                    } else {
                        const cx rr = buf[((j & 1) ? sb1b : sb1a) + kWave * j];
                        v[0] = rr.y; v[1] = rr.x;
                    }
                    const int t0 = 2 * C1 * (kWave * j + lane) - shift;
#pragma unroll
                    for (int i = 0; i < 2 * C1; ++i)
                        if (t0 + i >= 0 && t0 + i < nt) acc += v[i] * v[i];
                }
                acc = wave_reduce_add(acc);
                if (lane == 0) energy[(size_t)(b - 1) * npix + p] = acc;
                wave_sync();
                if (b == n_bands) break;
            }
            ad.refresh();
            float hs[R1][2 * C1];
            f_load_spec<PL>(H + (size_t)b * nf, lane, hs);
            const float h_nyq = H[(size_t)b * nf + N].x;
#pragma unroll
            for (int j = 0; j < R1; ++j) {
                if constexpr (C1 == 2) {
                    const cx e0 = cx_mul(cx{xs[j][0], xs[j][1]}, cx{hs[j][0], hs[j][1]});
                    const cx e1 = cx_mul(cx{xs[j][2], xs[j][3]}, cx{hs[j][2], hs[j][3]});
                    st2(buf + sb2 + 2 * kWave * j, swap2 ? e1 : e0, swap2 ? e0 : e1);
                } else {
                    buf[((j & 1) ? sb1b : sb1a) + kWave * j] = cx_mul(cx{xs[j][0], xs[j][1]}, cx{hs[j][0], hs[j][1]});
                }
            }
            if (lane == 0) buf[N] = cx{x_nyq * h_nyq, 0.0f};
            wave_sync();
            cx r[C1][R1];
            f_inverse_input<PL, false>(buf, w2n_s, wg_s, nullptr, lane, r);
            wave_sync();
            f_core_pass1<PL>(r, buf, t1, ad, lane);
            f_core_pass23<PL>(buf, t2, ad, lane);
        }
    }
}

// Round 3 (second half): the band energies WITHOUT a transform per band and per 2 M samples.  The "same" slice is the
// full linear convolution (nt + 2 s samples, s = shift = (taps - 1) / 2; no wrap in the M-point circular one) minus its
// first s and last s samples, so
//     E_b = sum_n y_b[n]^2  -  sum_(n < s) y_b[n]^2  -  sum_(n >= s + nt) y_b[n]^2
// * the first sum is Parseval's: (1 / M) sum_k |X[k] M H_b[k]|^2 = sum_(k <= M/2) |X[k]|^2 G_b[k],
//   G_b[k] = c_k M |H_b[k]|^2 (c = 1 at DC and Nyquist, 2 between) — seventeen multiply-adds per lane and band;
// * the head y_b[0 .. s) depends only on x[0 .. s) and h_b[0 .. s), the tail only on x[nt - s .. nt) and
//   h_b[s + 1 .. 2 s]: two linear convolutions of s x s samples (2 s - 1 <= 512), both real — so ONE complex 512-point
//   inverse transform per band gives both, head in the real part and tail in the imaginary part:
//       z = xh + i xt,  Z = FFT_512(z)  (once per pixel),   W_b[k] = Z[k] Hp_b[k] + conj(Z[-k]) Hm_b[k],
//       Hp = (Hh + Ht) / 2, Hm = (Hh - Ht) / 2   (Hh / Ht: 512-point spectra of the two filter halves, / 512)
//       w = IFFT(W_b):  head[n] = Re w[n], n < s;   tail[r] = Im w[s - 1 + r], r < s.
// A 512-point complex transform on the FPlan1024 core instead of a 1024-point one with its C2R merge (M = 2048): about
// a third of the instructions per band, a quarter of the LDS traffic, 4 KiB of LDS per wave.  A band's same-slice
// energy is at least about half of its full energy for a pulse anywhere in the trace (the filters are symmetric about
// tap s, which always falls inside the slice), so the subtraction costs at most about one bit.
struct DcPvDev {
    const cx *t1, *t2;  // FPlan1024 core tables (complex length 512)
    const cx2 *hpm;     // [n_bands][512] {Hp, Hm}
    const float *g;     // [n_bands][gstride], zero beyond bin M / 2
    int gstride;
};

// g[b][k] = c_k M |H_b[k]|^2 (k < nk; 0 up to gstride), hpm[b][k] = {(Hh + Ht) / 2, (Hh - Ht) / 2}; hht: [2 b] = Hh_b, [2 b + 1] = Ht_b
__global__ __launch_bounds__(256) void k_dc_pv_tables(int n_bands, int nk, int gstride, float M, const c32 *__restrict__ H,
                                                      const c32 *__restrict__ hht, float *__restrict__ g,
                                                      cx2 *__restrict__ hpm)
{
    const int total_g = n_bands * gstride, total_h = n_bands * 512;
    for (int e = (int)(blockIdx.x * blockDim.x + threadIdx.x); e < total_g + total_h; e += (int)(gridDim.x * blockDim.x)) {
        if (e < total_g) {
            const int b = e / gstride, k = e % gstride;
            float v = 0.0f;
            if (k < nk) {
                const c32 h = H[(size_t)b * nk + k];
                const double m2 = (double)h.re * (double)h.re + (double)h.im * (double)h.im;
                v = (float)(((k == 0 || k == nk - 1) ? 1.0 : 2.0) * (double)M * m2);
            }
            g[e] = v;
        } else {
            const int f = e - total_g, b = f / 512, k = f % 512;
            const c32 hh = hht[(size_t)(2 * b) * 512 + k], ht = hht[(size_t)(2 * b + 1) * 512 + k];
            hpm[f] = cx2{cx{0.5f * (hh.re + ht.re), 0.5f * (hh.im + ht.im)}, cx{0.5f * (hh.re - ht.re), 0.5f * (hh.im - ht.im)}};
        }
    }
}

// Two launches.  k_dc_energy_full: the Parseval sums — PX pixels per wave (their |X|^2 in registers, bins 256 i + 4 lane
// + c, the Nyquist bin apart), every band's G row read once per PX pixels; bound by reading the spectra.
// k_dc_energy_edges: one pixel per wave, eight waves per block; the block walks the bands together and stages each
// band's {Hp, Hm} row (8 KiB) in LDS one band ahead — read from L2 by every wave the rows were 80 GB per call at
// 512 x 512 pixels and the kernel waited for them 83 % of its time (first build) — and subtracts the edges' energy
// from what the first launch stored.
template <int NG, int PX>
__global__ __launch_bounds__(256) void k_dc_energy_full(DcPvDev T, size_t npix, int n_bands, int nk,
                                                        const cx *__restrict__ spec, float *__restrict__ energy)
{
    const int lane = lane_id();
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const size_t n_waves = (size_t)gridDim.x * (blockDim.x >> 6);
    for (size_t p0 = wave * PX; p0 < npix; p0 += n_waves * PX) {
        float pw[PX][NG][4], pn[PX];
#pragma unroll
        for (int q = 0; q < PX; ++q) {
            const bool on = p0 + q < npix;
            const cx *xs = spec + (on ? p0 + q : p0) * (size_t)nk;
#pragma unroll
            for (int i = 0; i < NG; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) {  // rows of nk = 256 NG + 1 bins start 8-byte aligned only
                    const cx u = xs[256 * i + 4 * lane + c];
                    pw[q][i][c] = on ? u.x * u.x + u.y * u.y : 0.0f;
                }
            const cx xn = xs[256 * NG];
            pn[q] = (on && lane == 0) ? xn.x * xn.x + xn.y * xn.y : 0.0f;
        }
#pragma unroll 1
        for (int b = 0; b < n_bands; ++b) {
            const float *gt = T.g + (size_t)b * T.gstride;
            constexpr int GB = NG < 8 ? NG : 8;  // rows of G in flight at a time
            const float gn = gt[256 * NG];
            float mine = 0.0f, acc[PX];
#pragma unroll
            for (int q = 0; q < PX; ++q) acc[q] = pn[q] * gn;
#pragma unroll
            for (int i0 = 0; i0 < NG; i0 += GB) {
                float4 g4[GB];
#pragma unroll
                for (int i = 0; i < GB; ++i) g4[i] = *reinterpret_cast<const float4 *>(gt + 256 * (i0 + i) + 4 * lane);
#pragma unroll
                for (int q = 0; q < PX; ++q)
#pragma unroll
                    for (int i = 0; i < GB; ++i) {
                        acc[q] += pw[q][i0 + i][0] * g4[i].x;
                        acc[q] += pw[q][i0 + i][1] * g4[i].y;
                        acc[q] += pw[q][i0 + i][2] * g4[i].z;
                        acc[q] += pw[q][i0 + i][3] * g4[i].w;
                    }
            }
#pragma unroll
            for (int q = 0; q < PX; ++q) {
                const float a = wave_reduce_add(acc[q]);
                if (lane == q) mine = a;
            }
            if (lane < PX && p0 + lane < npix) energy[(size_t)b * npix + p0 + lane] = mine;
        }
    }
}

template <int W>
struct DcPvLds {
    using PL = FPlan1024;
    static constexpr int TAB_ENTRIES = 2 * PL::N;                       // cx units: a band's 512 {Hp, Hm}
    static constexpr int TAB_UNITS = TAB_ENTRIES / 2;                   // 16-byte units
    static constexpr int UNITS = TAB_UNITS / (W * 64);                  // ... per thread
    static constexpr int TAB_OFF = PL::T1_ENTRIES + PL::T2_ENTRIES + W * PL::WAVE_ENTRIES;
    static constexpr size_t bytes() { return (size_t)(TAB_OFF + 2 * TAB_ENTRIES) * sizeof(cx); }
    static_assert(TAB_OFF % 2 == 0 && UNITS * W * 64 == TAB_UNITS, "16-byte units, the same number for every thread");
};

// SHIFT: the edges' length as a compile-time number (249 for the reference's 499 taps: which of a lane's eight output
// samples are head, tail or neither is then known per j, only j = 3 and j = 7 keep a lane test), or 0: run-time `shift`
// W: waves (= pixels) per block, 8 or 4
template <int SHIFT, int W>
__global__ __launch_bounds__(W * 64) THZ_WAVES_PER_SIMD(4) void k_dc_energy_edges(DcPvDev T, size_t npix, int nt,
                                                                                  int n_bands, int shift_rt,
                                                                                  const float *__restrict__ in,
                                                                                  float *__restrict__ energy)
{
    using PL = FPlan1024;
    using L = DcPvLds<W>;
    THZ_DYN_LDS(lds);
    constexpr int R1 = PL::R1, NC = PL::N, U = L::UNITS;
    static_assert(PL::C1 == 1 && NC == 512 && R1 == 8, "edge transform: 512 complex points, one column per lane");
    const int shift = SHIFT > 0 ? SHIFT : shift_rt;
    const int lane = lane_id();
    const int tid = (int)threadIdx.x;
    const int wib = THZ_UNIFORM((int)(threadIdx.x >> 6));
    cx *t1 = reinterpret_cast<cx *>(lds);
    cx *t2 = t1 + PL::T1_ENTRIES;
    cx *buf = t2 + PL::T2_ENTRIES + (size_t)wib * PL::WAVE_ENTRIES;
    cx *tab = t1 + L::TAB_OFF;
    for (int i = tid; i < PL::T1_ENTRIES; i += W * kWave) t1[i] = T.t1[i];
    for (int i = tid; i < PL::T2_ENTRIES; i += W * kWave) t2[i] = T.t2[i];
    const float4 *hsrc = reinterpret_cast<const float4 *>(T.hpm) + tid;  // a band's row: 512 16-byte units, U per thread
    float4 *hdst = reinterpret_cast<float4 *>(tab) + tid;
    float4 stage[U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
        stage[i] = hsrc[W * kWave * i];
        hdst[W * kWave * i] = stage[i];
    }
    FAddr<PL> ad;
    ad.init(lane);
    const int sba = nat(lane), sbb = nat(kWave + lane) - kWave;  // nat(64 j + lane) - 64 j for even / odd j
    const size_t n_batches = (npix + W - 1) / W;
    cx tw1[R1];  // W_512^(lane k1)
#pragma unroll
    for (int k1 = 1; k1 < R1; ++k1) tw1[k1] = T.t1[k1 * PL::M1 + lane];
    tw1[0] = cx{1.0f, 0.0f};
    cx za[R1], zb[R1];  // Z[k], conj Z[512 - k] at k = 64 j + lane
    bool edges_nz = true;
    unsigned t = 0;
    for (size_t q = blockIdx.x; q < n_batches; q += gridDim.x) {
        const size_t p = q * W + (size_t)wib;
        const bool live = p < npix;  // wave-uniform
        const bool last_batch = q + gridDim.x >= n_batches;
#pragma unroll 1
        for (int b = 0; b < n_bands; ++b, ++t) {
            __syncthreads();  // this band's row is in tab[t & 1]; nobody reads the other half any more
            const bool more = !(last_batch && b + 1 == n_bands);
            if (more) {
#pragma unroll
                for (int i = 0; i < U; ++i) stage[i] = hsrc[(size_t)(b + 1 < n_bands ? b + 1 : 0) * NC + W * kWave * i];
            }
            const cx *tb = tab + (size_t)(t & 1u) * L::TAB_ENTRIES;
            if (live) {
                float full = 0.0f;
                if (lane == 0 && (b == 0 || edges_nz)) full = energy[(size_t)b * npix + p];  // what k_dc_energy_full stored
                if (b == 0) {
                    // ---- Z = FFT_512(xh + i xt): xh[n] = x[n], xt[n] = x[nt - s + n], n < s
                    const float *x = in + p * (size_t)nt;
                    ad.refresh();
                    cx r[1][R1];
                    float nzf = 0.0f;
#pragma unroll
                    for (int j = 0; j < R1; ++j) {
                        const int n = kWave * j + lane, m = nt - shift + n;
                        const float xa = (n < shift && n < nt) ? x[n] : 0.0f;
                        const float xb = (n < shift && m >= 0) ? x[m] : 0.0f;
                        r[0][j] = cx{xa, xb};
                        nzf += (xa != 0.0f || xb != 0.0f) ? 1.0f : 0.0f;
                    }
                    // a trace whose first and last `shift` samples are all zero (a scan behind a time band pass narrower
                    // than the trace) has no edges to subtract: every band's transform would return exact zeros
                    edges_nz = wave_reduce_add(nzf) > 0.0f;  // wave-uniform
                    if (edges_nz) {
                        f_core_pass1<PL>(r, buf, t1, ad, lane);
                        f_core_pass23<PL>(buf, t2, ad, lane);
#pragma unroll
                        for (int j = 0; j < R1; ++j) {
                            za[j] = buf[((j & 1) ? sbb : sba) + kWave * j];
                            zb[j] = cx_conj(buf[nat(NC - kWave * j - lane)]);
                        }
                        wave_sync();
                    }
                }
                if (edges_nz) {
                ad.refresh();
                cx r[1][R1];
                const cx *hb = tb + 2 * launder_v(lane);
#pragma unroll
                for (int j = 0; j < R1; ++j) {
                    const cx2 h = ld2(hb + 2 * kWave * j);
                    const cx w = cx_mul_pk(za[j], h.a) + cx_mul_pk(zb[j], h.b);
                    r[0][j] = cx{w.y, w.x};  // swapped in, swapped out: the inverse transform through the forward core
                }
                // the lane's pass-1 twiddles stay in registers across the bands, and the outputs never go to LDS: of the
                // 39 KiB a band moved through LDS (the kernel's bound beside its VALU work) 11.5 less
                cx outv[R1];
                f_core_pass1_regs<PL>(r, buf, tw1, ad);
                f_core_pass23<PL, true, true>(buf, t2, ad, lane, &outv);
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < R1; ++j) {
                    const cx v = outv[j];  // {Im w[n], Re w[n]}, n = 64 j + lane
                    const int n = kWave * j + lane;
                    // with SHIFT known the j-th sample's range [64 j, 64 j + 63] settles most of these at compile time
                    const bool all_h = SHIFT > 0 && kWave * j + kWave - 1 < SHIFT, no_h = SHIFT > 0 && kWave * j >= SHIFT;
                    const bool all_t = SHIFT > 0 && kWave * j >= SHIFT - 1 && kWave * j + kWave - 1 < 2 * SHIFT - 1;
                    const bool no_t = SHIFT > 0 && (kWave * j + kWave - 1 < SHIFT - 1 || kWave * j >= 2 * SHIFT - 1);
                    if (!no_h) {
                        const float hd = (all_h || n < shift) ? v.y : 0.0f;
                        acc += hd * hd;
                    }
                    if (!no_t) {
                        const float tl = (all_t || (n >= shift - 1 && n < 2 * shift - 1)) ? v.x : 0.0f;
                        acc += tl * tl;
                    }
                }
                acc = wave_reduce_add(acc);
                if (lane == 0) energy[(size_t)b * npix + p] = full - acc;
                }
            }
            if (more) {
#pragma unroll
                for (int i = 0; i < U; ++i) hdst[(size_t)((t + 1u) & 1u) * L::TAB_UNITS + W * kWave * i] = stage[i];
            }
        }
    }
}

// Padded lengths without an F core (M = 8192: traces of 3599 .. 7694 samples): the recombination's multiplier sum as a
// kernel of its own.  In k_dc_combine every pixel reads all n_bands rows of H from L2 (25 x 32 KiB at M = 8192) at one or
// two waves per CU — 59 of the 79 ms of the call at 256 x 256 x 4000.  Here a block keeps a 256-bin slice of every
// band's row in LDS and walks the pixels: y[p][k] = X[p][k] sum_b g_b[p] H_b[k]; k_dc_combine then only transforms
// (n_bands = 0).
constexpr int kDcWeightBins = 256;
__global__ __launch_bounds__(256) void k_dc_weight_spectra(size_t npix, size_t gain_stride, int nk, int n_bands, unsigned gx,
                                                           const cx *__restrict__ spec, const cx *__restrict__ H,
                                                           const float *__restrict__ gain, cx *__restrict__ y)
{
    THZ_DYN_LDS(lds);
    cx *h_s = reinterpret_cast<cx *>(lds);  // [n_bands][256]
    // block -> (slice of bins, group of pixels): gx slices, gridDim.x / gx groups
    const unsigned slice = blockIdx.x % gx, group = blockIdx.x / gx, groups = gridDim.x / gx;
    const int k0 = (int)slice * kDcWeightBins;
    for (int i = (int)threadIdx.x; i < n_bands * kDcWeightBins; i += (int)blockDim.x) {
        const int b = i / kDcWeightBins, k = k0 + i % kDcWeightBins;
        h_s[i] = k < nk ? H[(size_t)b * nk + k] : cx{0.0f, 0.0f};
    }
    __syncthreads();
    const int lane = lane_id();
    const int wib = (int)(threadIdx.x >> 6), wpb = (int)(blockDim.x >> 6);
    const int kl = k0 + 4 * lane;
    for (size_t p = (size_t)group * wpb + wib; p < npix; p += (size_t)groups * wpb) {
        cx x[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) x[c] = kl + c < nk ? spec[p * (size_t)nk + kl + c] : cx{0.0f, 0.0f};
        cx hc[4] = {cx{0.0f, 0.0f}, cx{0.0f, 0.0f}, cx{0.0f, 0.0f}, cx{0.0f, 0.0f}};
        const cx *hl = h_s + 4 * launder_v(lane);
        for (int b = 0; b < n_bands; ++b) {
            const float g = gain[(size_t)b * gain_stride + p];
            const cx2 h01 = ld2(hl + b * kDcWeightBins), h23 = ld2(hl + b * kDcWeightBins + 2);
            hc[0] += cx{g, g} * h01.a;
            hc[1] += cx{g, g} * h01.b;
            hc[2] += cx{g, g} * h23.a;
            hc[3] += cx{g, g} * h23.b;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (kl + c < nk) y[p * (size_t)nk + kl + c] = cx_mul(x[c], hc[c]);
    }
}

__global__ __launch_bounds__(256) void k_dc_combine(PlanDev P, size_t npix, int nt, int n_bands,
                                                    int shift, const c32 *__restrict__ spec,
                                                    const c32 *__restrict__ H,
                                                    const float *__restrict__ gain,
                                                    float *__restrict__ out,
                                                    float *__restrict__ img)
{
    THZ_DYN_LDS(lds);
    const int lane = lane_id();
    const int wib = (int)(threadIdx.x >> 6);
    const int wpb = (int)(blockDim.x >> 6);
    const int N = 1 << P.log2n;
    // two buffers per wave; n_bands == 0: `spec` already holds X . sum_b g_b H_b (k_dc_weight_spectra), H and gain unused
    c32 *A = reinterpret_cast<c32 *>(lds + (size_t)wib * (size_t)(2 * P.buf_entries) * sizeof(c32));
    c32 *B = A + P.buf_entries;
    for (size_t p = (size_t)blockIdx.x * wpb + wib; p < npix; p += (size_t)gridDim.x * wpb) {
        for (int k = lane; k <= N; k += kWave) {
            if (n_bands == 0) {  // wave-uniform
                A[k] = spec[p * (size_t)(N + 1) + k];
                continue;
            }
            c32 hc = c32{0.0f, 0.0f};
            for (int b = 0; b < n_bands; ++b) {
                const float g = gain[(size_t)b * npix + p];
                const c32 h = H[(size_t)b * (N + 1) + k];
                hc.re += g * h.re;
                hc.im += g * h.im;
            }
            A[k] = cmul(spec[p * (size_t)(N + 1) + k], hc);
        }
        wave_sync();
        c2r_merge_swapped(A, B, N, P.tw_split, lane);
        const c32 *R = wave_cfft(B, A, P.log2n, P.tw, lane);
        float acc = 0.0f;
        for (int t = lane; t < nt; t += kWave) {
            const float v = dc_real_at(R, t + shift);
            out[p * nt + t] = v;
            acc += v * v;
        }
        if (img) {
            acc = wave_reduce_add(acc);
            if (lane == 0) img[p] = acc;
        }
        wave_sync();
    }
}

// ---- Richardson–Lucy, batched over bands -----------------------------------
// index of the first block of band b in the flattened grid is blk0[b]
__device__ __forceinline__ int rl_find_band(const RlBand *bands, int n_bands, unsigned blk,
                                            unsigned *local_blk)
{
    int b = 0;
    while (b + 1 < n_bands && blk >= bands[b + 1].blk0) ++b;
    *local_blk = blk - bands[b].blk0;
    return b;
}

// reflect padding of the energy image, deconvolution.rs:629-670; u = d
__global__ __launch_bounds__(256) void k_rl_init(const RlBand *__restrict__ bands, int n_bands,
                                                 size_t npix, const float *__restrict__ energy,
                                                 float *__restrict__ ws)
{
    unsigned lb;
    const int b = rl_find_band(bands, n_bands, blockIdx.x, &lb);
    const RlBand B = bands[b];
    const int idx = (int)(lb * blockDim.x + threadIdx.x);
    if (idx >= B.H * B.W) return;
    const int Y = idx / B.W, X = idx % B.W;
    // rows first (from the image), then columns (from the row-padded array)
    int xs = X - B.pad_x;                 // column in image coordinates
    if (X < B.pad_x) xs = B.pad_x - X;              // src col pad_x + (pad_x - j), j = X
    else if (X >= B.pad_x + B.w) xs = B.w - 2 - (X - B.pad_x - B.w);
    int ys = Y - B.pad_y;
    if (Y < B.pad_y) ys = B.pad_y - Y;
    else if (Y >= B.pad_y + B.h) ys = B.h - 2 - (Y - B.pad_y - B.h);
    const float v = energy[(size_t)b * npix + (size_t)ys * B.w + xs];
    ws[B.off_d + idx] = v;
    ws[B.off_u + idx] = v;
}

// one "same" 2-D convolution value, deconvolution.rs:432-458 (mode 0: kernels of
// <= 256 elements, correlation-indexed) or the FFT path's intended result
// (mode 1: true convolution, offset (b-1)/2).  m outer / n inner, no FMA
// contraction, like the CPU loops.
__device__ __forceinline__ float rl_conv_at(const float *__restrict__ a, int H, int W,
                                            const float *__restrict__ k, int pr, int pc, int mode,
                                            int i, int j)
{
#pragma clang fp contract(off)
    float sum = 0.0f;
    if (mode == 0) {
        const int hr = pr / 2, hc = pc / 2;
        for (int m = 0; m < pr; ++m) {
            const int x = i + m - hr;
            if (x < 0 || x >= H) continue;
            for (int n = 0; n < pc; ++n) {
                const int y = j + n - hc;
                if (y >= 0 && y < W) sum += a[(size_t)x * W + y] * k[m * pc + n];
            }
        }
    } else {
        const int sr = (pr - 1) / 2, sc = (pc - 1) / 2;
        for (int m = 0; m < pr; ++m) {
            const int x = i + sr - m;
            if (x < 0 || x >= H) continue;
            for (int n = 0; n < pc; ++n) {
                const int y = j + sc - n;
                if (y >= 0 && y < W) sum += a[(size_t)x * W + y] * k[m * pc + n];
            }
        }
    }
    return sum;
}

// step 1: t = d / (u (*) psf + eps)     step 2: u *= t (*) mirror(psf)
__global__ __launch_bounds__(256) void k_rl_step(const RlBand *__restrict__ bands, int n_bands,
                                                 const int *__restrict__ it_base, int iteration, int step,
                                                 float *__restrict__ ws)
{
#pragma clang fp contract(off)
    if (it_base) iteration += *it_base;  // graph replay: the batch's first iteration lives in memory
    unsigned lb;
    const int b = rl_find_band(bands, n_bands, blockIdx.x, &lb);
    const RlBand B = bands[b];
    if (iteration >= B.n_iter) return;
    const int idx = (int)(lb * blockDim.x + threadIdx.x);
    if (idx >= B.H * B.W) return;
    const int i = idx / B.W, j = idx % B.W;
    if (step == 0) {
        const float c = rl_conv_at(ws + B.off_u, B.H, B.W, ws + B.off_psf, B.pr, B.pc, B.mode, i, j);
        ws[B.off_t + idx] = ws[B.off_d + idx] / (c + 1e-12f);
    } else {
        const float c = rl_conv_at(ws + B.off_t, B.H, B.W, ws + B.off_mirror, B.pr, B.pc, B.mode, i, j);
        ws[B.off_u + idx] = ws[B.off_u + idx] * c;
    }
}

// The same step with the image tile and the PSF in LDS: a block owns a 16 x 16 tile of one band's
// padded image, stages the tile plus its halo (zeros outside the image — the reference skips
// those taps, and sum + 0*k leaves the sum as it was) and the kernel, then every thread runs the
// reference's m-outer / n-inner loop out of LDS.  Values are those of k_rl_step bit for bit
// (same operands, same order, no FMA); it only stops fetching every tap from L2.
constexpr int kRlTile = 16;

// One pixel's sum over the pr x pc taps in the reference's order (m outer, n inner, one add per tap,
// no FMA), out of LDS.  The sum is a single dependent chain.  Most iterations only the two or three
// widest bands are still running (n_iter falls with frequency), which is about one wave per SIMD:
// nothing else hides a wait, and a launch lasts exactly as long as one such wave's chain.  What made
// that slow was never the adds but the operand latency in front of each group of them — taps
// fetched through the scalar cache (cold at every launch: an L2 round trip per 64 bytes) or LDS
// reads waited for in full before the first multiply, per 16 taps and per TAP in a row's remainder.
// Here a row is cut into chunks of kRlChunk taps, the chunks of all rows form one sequence, the
// taps are staged into LDS IN THAT SEQUENCE (one coalesced pass per block), and chunk t+1's
// operands are requested before chunk t's arithmetic starts; LDS returns in order, so the wait in
// front of a chunk leaves the next chunk's reads in flight.  Every fetch is a full chunk: a row's
// last chunk is the row's LAST kRlChunk taps (it overlaps the chunk before it, or starts in front
// of the row when the row is shorter than a chunk) and the positions in front of the row's own taps
// are staged as ZERO taps: sum + a * 0 is sum, bit for bit.  The image-side surplus is read from the
// neighbouring row of the tile or from the slack the tile carries on either side (zeroed: 0 * stale
// LDS could be a NaN).
constexpr int kRlChunk = 16;

// chunks per row / first tap of chunk c of a row
__device__ __forceinline__ int rl_chunks(int pc) { return (pc + kRlChunk - 1) / kRlChunk; }
__device__ __forceinline__ int rl_chunk_n0(int c, int nch, int pc) { return c + 1 == nch ? pc - kRlChunk : c * kRlChunk; }

__device__ __forceinline__ float rl_tile_taps(const float *origin, int wsz, const float *k_s, int pr, int pc)
{
#pragma clang fp contract(off)
    // A chunk's 16 taps are held one per lane (lane l of every row of 16 lanes has tap l) and reach the products
    // through DPP row broadcasts inside the multiplies (row_mul16): one 4-byte LDS read per lane and chunk instead of
    // sixteen dwords broadcast to every lane — the tap loop was bound by the LDS port (72 clocks per tap with 16
    // waves on a CU), half of it these broadcasts.  Every lane of the wave must call this (the edge tiles' lanes
    // outside the image included: they hold taps for their row).
    const int nch = rl_chunks(pc);
    const int total = pr * nch;
    const int lane16 = (int)(threadIdx.x & 15);
    int f_a = 0, f_c = 0, f_m = 0, f_t = 0;  // fetch position: row start in the tile, chunk of the row, row, chunk of the sequence
    auto fetch = [&](float (&av)[kRlChunk], float &kl) {
        const float *row = origin + (f_a + rl_chunk_n0(f_c, nch, pc));
        kl = k_s[f_t * kRlChunk + lane16];  // first: LDS returns in order, and the tap is what a chunk's first multiply waits for
#pragma unroll
        for (int q = 0; q < kRlChunk; ++q) av[q] = row[q];
        ++f_t;
        ++f_c;
        if (f_c == nch) {
            f_c = 0;
            if (f_m + 1 < pr) {  // the fetches after the last chunk repeat the last row (their taps are zeros)
                ++f_m;
                f_a += wsz;
            }
        }
    };
    float sum = 0.0f;
    // Positions that are not taps — the front of a row's end-aligned last chunk, which belongs to the chunk before,
    // and an odd sequence's filler chunk — are staged as zeros: sum + a * 0 is sum, bit for bit (a is finite: the
    // iteration divides by sums + 1e-12), so the loop has no cases and every broadcast has one consumer to fold into.
    auto accumulate = [&](const float (&av)[kRlChunk], float kl) {
        float prod[kRlChunk];
        row_mul16(av, kl, prod);
#pragma unroll
        for (int q = 0; q < kRlChunk; ++q) sum += prod[q];
    };
    float a0[kRlChunk], a1[kRlChunk], k0, k1;
    fetch(a0, k0);
    for (int t = 0; t < total; t += 2) {
        fetch(a1, k1);
        accumulate(a0, k0);
        fetch(a0, k0);
        accumulate(a1, k1);
    }
    return sum;
}

// Wide kernels (mode 1, more than 256 taps) stand for the reference's FFT convolution, whose
// rounding is not that of any particular summation order, so their sums need not be one chain.
// Late in a call only the widest bands still iterate — one tile per CU — and a launch then costs
// what one CU needs for one tile.  With a thread per pixel that is LDS bandwidth: every tap moves
// 4 bytes of image and 4 bytes of kernel per lane through the 128 bytes/clock LDS port (measured:
// 35 us per launch, whichever way the operands were fetched).  So a thread owns FOUR pixels side
// by side and reads a sliding window — 20 image values and 16 taps (a broadcast) feed 64 FMAs — and
// the rows of the kernel are dealt to kRlSplit waves, whose partial sums meet in LDS.
// The tile is stored turned by 180 degrees (mode 1 walks the image downwards in both axes):
// a[(ti + pr-1 - m), (tj + pc-1 - n)] is turned[(15-ti) + m][(15-tj) + n].  Its row stride is padded
// so that the 16-byte reads of a quarter wave (16 tile rows, one column group) fall into different banks.  Taps are
// staged row by row, each row zero-padded to whole chunks; everything a window can reach beyond a
// row's taps is initialised (0 * NaN from stale LDS would poison the sum).
constexpr int kRlSplit = 16;            // waves per tile
constexpr int kRlThreads = 64 * kRlSplit;
constexpr int kRlPix = 4;               // pixels per thread
constexpr int kRlTilesPerBlock = 1;     // narrow-kernel tiles per block (one per 256 threads).  Four (round 2's first
                                        // arrangement: the taps staged once for four tiles) put four waves on every
                                        // SIMD of a CU while other CUs idled: a launch lasts as long as its fullest CU
constexpr int kRlNarrowThreads = 256 * kRlTilesPerBlock;

__host__ __device__ inline int rl_turned_stride(int wsz)
{
    // a window reaches up to kRlChunk columns past the halo (the zero-padded taps); a quarter wave
    // reads 16 bytes in each of 16 consecutive tile rows, which hit 16 different bank groups when
    // the stride is an odd number of 16-byte units
    int w = (wsz + kRlChunk + 3) / 4 * 4;
    if (w / 4 % 2 == 0) w += 4;
    return w;
}

typedef float rl_f2 __attribute__((ext_vector_type(2)));

// Pixels in pairs, taps one at a time: (sum[p+1], sum[p]) += (w[x], w[x+1]) * (k, k) is one packed FMA when
// (w[x], w[x+1]) is an even-aligned register pair — which it is for every other tap.  The odd pairs
// are made in registers (one v_pk_mov_b32 each) rather than read from LDS a second time.
__device__ __forceinline__ void rl_tile_taps_split(const float *window0, int wsp, const float *k_s, int pc,
                                                   int m_begin, int m_end, float (&acc)[kRlPix])
{
    static_assert(kRlPix == 4 && kRlChunk == 16, "register layout below");
    const int nch = rl_chunks(pc);
    rl_f2 s10 = {0.0f, 0.0f}, s32 = {0.0f, 0.0f};  // (pixel 1, pixel 0), (pixel 3, pixel 2)
    for (int m = m_begin; m < m_end; ++m) {
        const float4 *row = reinterpret_cast<const float4 *>(window0 + m * wsp);
        const float4 *kr = reinterpret_cast<const float4 *>(k_s + m * nch * kRlChunk);
        for (int c = 0; c < nch; ++c) {
            rl_f2 we[10], wo[9];  // we[i] = (w[2i], w[2i+1]), wo[i] = (w[2i+1], w[2i+2])
            float kv[kRlChunk];
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const float4 v = row[c * 4 + q];
                we[2 * q] = rl_f2{v.x, v.y};
                we[2 * q + 1] = rl_f2{v.z, v.w};
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = kr[c * 4 + q];
                kv[4 * q] = v.x; kv[4 * q + 1] = v.y; kv[4 * q + 2] = v.z; kv[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) {
#ifdef THZ_EMU
                wo[i] = rl_f2{we[i].y, we[i + 1].x};
#else
                asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(wo[i]) : "v"(we[i]), "v"(we[i + 1]));
#endif
            }
            // pixel p, tap q reads w[3 - p + q]: pair (1, 0) reads (w[2 + q], w[3 + q]), pair (3, 2) reads (w[q], w[q + 1])
#pragma unroll
            for (int q = 0; q < kRlChunk; ++q) {
                const rl_f2 kk = {kv[q], kv[q]};
                const rl_f2 x10 = (q % 2 == 0) ? we[(2 + q) / 2] : wo[(1 + q) / 2];
                const rl_f2 x32 = (q % 2 == 0) ? we[q / 2] : wo[(q - 1) / 2];
                s10 = __builtin_elementwise_fma(x10, kk, s10);
                s32 = __builtin_elementwise_fma(x32, kk, s32);
            }
        }
    }
    acc[0] = s10.y; acc[1] = s10.x; acc[2] = s32.y; acc[3] = s32.x;
}

// LDS floats of a block: slack | tile + halo | slack | taps (chunk order, or padded rows) | partial sums
__host__ __device__ inline size_t rl_tile_floats(int pr, int pc, bool turned)
{
    const int wsz = kRlTile + pc - 1;
    const size_t tile = (size_t)(kRlTile + pr - 1) * (turned ? rl_turned_stride(wsz) : wsz);
    return (2 * (size_t)kRlChunk + tile + 3) / 4 * 4;  // taps start 16-byte aligned
}
__host__ __device__ inline size_t rl_tap_floats(int pr, int pc)
{
    return ((size_t)pr * ((pc + kRlChunk - 1) / kRlChunk) + 2) * kRlChunk;
}
__host__ __device__ inline bool rl_turned(int pr, int pc) { return pr * pc > 256; }  // RlBand::mode == 1

// A block of the tiled grid: one tile of a wide-kernel band (2-D sums: all 16 waves share it, see above), or
// kRlTilesPerBlock consecutive tiles of a narrow-kernel band, one per group of 256 threads — a narrow kernel keeps
// the reference's one-chain-per-pixel sums, so a tile can use 256 threads only.
__host__ __device__ inline unsigned rl_tile_blocks(bool turned, unsigned n_tiles)
{
    return turned ? n_tiles : (n_tiles + kRlTilesPerBlock - 1) / kRlTilesPerBlock;
}

// A block's band record and the batch's first iteration number, requested together as three scalar loads and
// waited for once (want_scalars_now): two round trips at the head of the kernel — the arguments, then everything
// else — where the compiler's field-by-field loads made six.
__device__ __forceinline__ RlBand rl_block_band(const RlTileRef *__restrict__ tiles, const int *__restrict__ it_base, int &iteration)
{
    const int *rec = reinterpret_cast<const int *>(tiles + blockIdx.x);
    const thz_i16 lo = *reinterpret_cast<const thz_i16 *>(rec);
    const thz_i8 hi = *reinterpret_cast<const thz_i8 *>(rec + 16);
    const int base = *(it_base ? it_base : rec);  // a load either way: no branch in front of the others
    want_scalars_now(lo, hi, base);
    struct { thz_i16 lo; thz_i8 hi; } both{lo, hi};
    RlBand B;
    __builtin_memcpy(&B, &both, sizeof(B));
    if (it_base) iteration += base;
    return B;
}

// The grid of a launch is a list of tiles of ONE kind (WIDE: kernels of more than 256 taps), the bands in it by
// falling iteration count, so that the tiles still iterating are a prefix of the list.  Two kernels rather than
// one with a branch: the wide path fits 64 VGPRs, which lets two 1024-thread blocks share a CU — one block's
// staging and partial-sum reduction then run under the other's tap loop — while the narrow path's double-buffered
// chunks need more registers and less LDS.
template <bool WIDE>
__global__ __launch_bounds__(WIDE ? kRlThreads : kRlNarrowThreads, WIDE ? 8 : 1) void k_rl_step_tiled(const RlTileRef *__restrict__ tiles,
                                                                            const int *__restrict__ it_base, int iteration,
                                                                            int step, float *__restrict__ ws)
{
    THZ_DYN_LDS(smem);
    // the tile's band record, by value in the per-tile table: the tiles of finished bands leave after a single
    // memory latency
    const RlBand B = rl_block_band(tiles, it_base, iteration);
    if (iteration >= B.n_iter) return;  // block-uniform
    const int pr = B.pr, pc = B.pc;
    const int hs = kRlTile + pr - 1, wsz = kRlTile + pc - 1;  // halo tile
    const int nch = rl_chunks(pc);
    const unsigned a_off = step == 0 ? B.off_u : B.off_t;
    const float *a = ws + a_off;
    const float *k = ws + (step == 0 ? B.off_psf : B.off_mirror);
    if constexpr (!WIDE) {
        // ---- narrow kernel: kRlTilesPerBlock tiles, one per 256-thread group, the taps staged once per block
        const int grp = (int)(threadIdx.x >> 8), px = (int)(threadIdx.x & 255);
        const unsigned lt = (blockIdx.x - B.tblk0) * kRlTilesPerBlock + (unsigned)grp;
        const bool live = lt < (unsigned)B.n_tiles;
        const int ti0 = (int)(lt / (unsigned)B.tiles_w) * kRlTile, tj0 = (int)(lt % (unsigned)B.tiles_w) * kRlTile;
        const size_t tile_fl = rl_tile_floats(pr, pc, false);
        float *a_s = reinterpret_cast<float *>(smem) + (size_t)grp * tile_fl + kRlChunk;  // slack in front and behind
        float *k_s = reinterpret_cast<float *>(smem) + kRlTilesPerBlock * tile_fl;
        // first image row / column of the halo: x = i + m - pr/2
        const int r0 = ti0 - pr / 2, c0 = tj0 - pc / 2;
        const int ti = px / kRlTile, tj = px % kRlTile;
        const int i = ti0 + ti, j = tj0 + tj;
        const bool writer = live && i < B.H && j < B.W;
        const int idx = i * B.W + j;
        // Everything is requested from memory before anything is waited for (the inputs were written by other CUs
        // one launch ago: every dependent load is two microseconds of a launch that computes for three): a tap in
        // the order rl_tile_taps consumes them, the update's other operand, then the halo, eight rows per wave at a time.
        const int n_el = (pr * nch + 2) * kRlChunk;  // the taps' chunks and the zeros of the filler chunks behind them
        const unsigned k_off = step == 0 ? B.off_psf : B.off_mirror;
        const int last_skip = nch * kRlChunk - pc;  // positions in front of a row's end-aligned last chunk: not taps
        auto tap_at = [&](int e) {
            const int t = e / kRlChunk, q = e % kRlChunk;
            const int m = t / nch, c = t - m * nch;
            const int src = m * pc + rl_chunk_n0(c, nch, pc) + q;
            const bool is_tap = m < pr && !(c + 1 == nch && q < last_skip);
            return ws[is_tap ? k_off + (unsigned)src : B.off_zero];
        };
        float other = 0.0f;
        const float k_v = tap_at((int)threadIdx.x), k_v1 = tap_at((int)threadIdx.x + kRlNarrowThreads);  // up to 30 chunks
        if (writer) other = ws[(step == 0 ? B.off_d : B.off_u) + idx];
        if (live) {  // zeros outside the image — the reference skips those taps, and sum + 0*k leaves the sum as it was
            constexpr int kRows = 8;
            const int wvu = THZ_UNIFORM(px >> 6), ln = px & 63;
            auto stage = [&](int rb, int c) {
                float v[kRows];
                const int y = c0 + c;
                const bool yin = y >= 0 && y < B.W;
#pragma unroll
                for (int q = 0; q < kRows; ++q) {
                    // rows are wave-uniform (scalar tests); a position outside the image loads a stored zero, so
                    // the loads are unconditional, issue back to back and leave no masks to keep
                    const int r = rb + 4 * q, x = r0 + r;
                    const bool row_ok = r < hs && x >= 0 && x < B.H;
                    v[q] = ws[row_ok && yin ? a_off + (unsigned)(x * B.W) + (unsigned)y : B.off_zero];
                }
#pragma unroll
                for (int q = 0; q < kRows; ++q) {
                    const int r = rb + 4 * q;
                    if (r < hs) a_s[r * wsz + c] = v[q];
                }
            };
            // the first batch (the whole halo for kernels of up to 17 x 49 taps) stands in front of the loops: at a
            // loop head the compiler waits for every load in flight, the tap's and the operand's included
            if (ln < wsz) stage(wvu, ln);
            for (int c = ln + kWave; c < wsz; c += kWave) stage(wvu, c);
            for (int rb = wvu + 4 * kRows; rb < hs; rb += 4 * kRows)
                for (int c = ln; c < wsz; c += kWave) stage(rb, c);
        }
        if (live) {  // the slack on either side of the tile meets zero taps only, but must be finite
            if (px < kRlChunk) a_s[px - kRlChunk] = 0.0f;
            if (px < (int)tile_fl - 2 * kRlChunk - hs * wsz + kRlChunk) a_s[hs * wsz + px] = 0.0f;
        }
        if ((int)threadIdx.x < n_el) k_s[threadIdx.x] = k_v;
        if ((int)threadIdx.x + kRlNarrowThreads < n_el) k_s[threadIdx.x + kRlNarrowThreads] = k_v1;
        for (int e = (int)threadIdx.x + 2 * kRlNarrowThreads; e < n_el; e += kRlNarrowThreads) k_s[e] = tap_at(e);
        __syncthreads();
        if (!live) return;  // whole waves; no barrier below on this path
        const float sum = rl_tile_taps(a_s + ti * wsz + tj, wsz, k_s, pr, pc);  // every lane: see there
        if (!writer) return;
        if (step == 0) ws[B.off_t + idx] = other / (sum + 1e-12f);
        else ws[B.off_u + idx] = other * sum;
        return;
    } else {
    // ---- wide kernel: one tile, stored turned by 180 degrees, the kernel's rows dealt to the 16 waves
    const unsigned lt = blockIdx.x - B.tblk0;
    const int ti0 = (int)(lt / (unsigned)B.tiles_w) * kRlTile, tj0 = (int)(lt % (unsigned)B.tiles_w) * kRlTile;
    const int wsp = rl_turned_stride(wsz);                    // the tile's row stride in LDS
    float *a_s = reinterpret_cast<float *>(smem) + kRlChunk;  // slack in front and behind
    float *k_s = reinterpret_cast<float *>(smem) + rl_tile_floats(pr, pc, true);
    float *part_s = k_s + rl_tap_floats(pr, pc);  // [kRlSplit][256]
    // first image row / column of the halo: x = i + (pr-1)/2 - m
    const int r0 = ti0 + (pr - 1) / 2 - (pr - 1), c0 = tj0 + (pc - 1) / 2 - (pc - 1);
    {   // a wave per halo row, lanes along the row, the block's loads all in flight together
        const int wv = (int)(threadIdx.x >> 6), ln = (int)(threadIdx.x & 63);
#pragma unroll 4
        for (int r = wv; r < hs; r += kRlThreads / kWave) {
            const int x = r0 + r;
            const bool xin = x >= 0 && x < B.H;
            float *dst = a_s + (hs - 1 - r) * wsp;
            for (int c = ln; c < wsp; c += kWave) {
                const int y = c0 + c;
                const float v = (c < wsz && xin && y >= 0 && y < B.W) ? a[(size_t)x * B.W + y] : 0.0f;
                dst[c < wsz ? wsz - 1 - c : c] = v;
            }
        }
        if (threadIdx.x < kRlChunk) a_s[hs * wsp + (int)threadIdx.x] = 0.0f;  // the slack behind the last row (the taps start right after it)
        // padded rows of taps: a wave per kernel row
        for (int m = wv; m < pr; m += kRlThreads / kWave)
            for (int n = ln; n < nch * kRlChunk; n += kWave) k_s[m * nch * kRlChunk + n] = n < pc ? k[m * pc + n] : 0.0f;
    }
    __syncthreads();
    const int px = (int)threadIdx.x;
    const int i = ti0 + px / kRlTile, j = tj0 + px % kRlTile;
    const bool writer = px < 256 && i < B.H && j < B.W;
    const int idx = i * B.W + j;
    // the other operand of the update does not depend on the sums: fetched now, it arrives under them
    float other = 0.0f;
    if (writer) other = ws[(step == 0 ? B.off_d : B.off_u) + idx];
    {
        // wave g sums kernel rows [g pr/16, (g+1) pr/16) for the whole tile: lane -> tile row and four columns
        const int g = (int)(threadIdx.x >> 6), ln = (int)(threadIdx.x & 63);
        const int ti = ln % kRlTile, tj = (ln / kRlTile) * kRlPix;
        float acc[kRlPix];
        rl_tile_taps_split(a_s + (kRlTile - 1 - ti) * wsp + (kRlTile - kRlPix - tj), wsp, k_s, pc,
                           g * pr / kRlSplit, (g + 1) * pr / kRlSplit, acc);
        *reinterpret_cast<float4 *>(part_s + g * 256 + ti * kRlTile + tj) = float4{acc[0], acc[1], acc[2], acc[3]};
    }
    __syncthreads();
    if (!writer) return;
    float sum = 0.0f;
#pragma unroll
    for (int g = 0; g < kRlSplit; ++g) sum += part_s[g * 256 + px];
    if (step == 0) ws[B.off_t + idx] = other / (sum + 1e-12f);
    else ws[B.off_u + idx] = other * sum;
    }
}

// ---- separable wide kernels ---------------------------------------------------------------------------------
// Every band PSF of the reference is an outer product of two 1-D profiles (create_psf_2d, psf.rs:228-313:
// psf[m][n] = fx[m] fy[n]), so the wide kernels' "same" convolution — in the reference an FFT convolution, whose
// rounding is that of no summation order — factors into a pass along the rows and a pass down the columns:
//     T[x][j]   = sum_n a[x][j + sc - n] fy[n]          (halo rows x, the tile's 16 columns)
//     out[i][j] = sum_m T[i + sr - m][j] fx[m]
// pr + pc multiply-adds per pixel and half-step instead of pr pc (104 against 2 679 for the 47 x 57 band), which
// turns a tile from ALU/LDS-bound into what its halo costs to fetch — hence tiles of kRlSepTileRows x kRlSepTileCols
// pixels (32 x 32: 1 024 threads, 42 KB of LDS for the largest band), which load 6.7 floats per pixel where 16 x 16
// load 20.  The halo is stored as it lies in the image and the
// profiles stored reversed, so both passes walk upwards: out[ti][tj] = sum_m' sum_n' a_s[ti + m'][tj + n']
// fx[pr-1-m'] fy[pc-1-n'].  The mirrored PSF of the second half-step has the profiles the other way round.
// Pass A reuses the wide kernel's window arithmetic (four pixels side by side per thread, packed FMAs): a
// quarter wave reads 16 consecutive halo rows at one column group, conflict-free for an odd row stride in 16-byte
// units.
static_assert(kRlSepTileCols * kRlSepTileRows == 1024, "the largest block of a separable tile: a thread per pixel");
constexpr int kRlSepQuads = kRlSepTileCols / kRlPix;              // column groups of four per halo row (pass A)
// halo rows a wave has in flight while staging: kernels of up to 49 rows in one batch (of at most twelve rows per wave)
constexpr int rl_sep_rows(int nt) { return (kRlSepTileRows + 48 + nt / 64 - 1) / (nt / 64) < 12 ? (kRlSepTileRows + 48 + nt / 64 - 1) / (nt / 64) : 12; }

__host__ __device__ inline int rl_sep_stride(int pc)
{
    // the last column group's window ends at column (tile width - 4) + 16 chunks + 3; everything up to the stride is
    // initialised
    int w = kRlChunk * ((pc + kRlChunk - 1) / kRlChunk) + kRlSepTileCols;
    if (w / 4 % 2 == 0) w += 4;
    return w;
}
// T of pass A is kept TRANSPOSED — column tj of the tile is the contiguous row tj of T', halo rows along it — so that
// pass B is pass A's window arithmetic turned by ninety degrees (four pixels below each other per thread, one 16-byte
// read per four halo rows and a broadcast of sixteen taps): the stride of a column is the column pass's counterpart of
// rl_sep_stride
__host__ __device__ inline int rl_sep_tstride(int pr) { return rl_sep_stride(pr); }
// LDS floats of a block: halo rows | fy (whole chunks) | fx (whole chunks) | T' (a row per tile column)
__host__ __device__ inline size_t rl_sep_floats(int pr, int pc)
{
    const int hs = kRlSepTileRows + pr - 1;
    return (size_t)hs * rl_sep_stride(pc) + (size_t)((pc + kRlChunk - 1) / kRlChunk) * kRlChunk
           + (size_t)((pr + kRlChunk - 1) / kRlChunk) * kRlChunk + (size_t)kRlSepTileCols * rl_sep_tstride(pr);
}

// NT threads per block (1 024, 512 or 256): a tile's passes need 624 / 256 tasks at most, and what a launch costs is the
// blocks' latency times the rounds of blocks the chip needs — fewer waves per block put more tiles on a CU at once
template <int NT>
__global__ __launch_bounds__(NT, 8) void k_rl_step_sep(const RlTileRef *__restrict__ tiles,
                                                               const int *__restrict__ it_base, int iteration, int step,
                                                               float *__restrict__ ws)
{
    THZ_DYN_LDS(smem);
    constexpr int kRows = rl_sep_rows(NT);
    const RlBand B = rl_block_band(tiles, it_base, iteration);
    if (iteration >= B.n_iter) return;  // block-uniform
    const int pr = B.pr, pc = B.pc;
    const int hs = kRlSepTileRows + pr - 1, wsz = kRlSepTileCols + pc - 1;
    const int nch = rl_chunks(pc), wsp = rl_sep_stride(pc);
    const int nchr = rl_chunks(pr), hsp = rl_sep_tstride(pr);  // the column pass: chunks of fx, stride of a column of T'
    const unsigned a_off = step == 0 ? B.off_u : B.off_t;
    const float *fx = ws + B.off_fx, *fy = ws + B.off_fy;
    float *a_s = reinterpret_cast<float *>(smem);
    float *fy_s = a_s + (size_t)hs * wsp;
    float *fx_s = fy_s + nch * kRlChunk;
    float *t_s = fx_s + nchr * kRlChunk;  // T'[tj][r], r < hsp
    const unsigned lt = blockIdx.x - B.tblk0;
    const int ti0 = (int)(lt / (unsigned)B.tiles_w) * kRlSepTileRows, tj0 = (int)(lt % (unsigned)B.tiles_w) * kRlSepTileCols;
    // first image row / column of the halo: x = i + (pr-1)/2 - m
    const int r0 = ti0 + (pr - 1) / 2 - (pr - 1), c0 = tj0 + (pc - 1) / 2 - (pc - 1);
    const int wv = (int)(threadIdx.x >> 6), ln = (int)(threadIdx.x & 63);
    const int px = (int)threadIdx.x;
    // pass B and the update: the first 256 threads, each four pixels below each other in one column of the tile
    constexpr int kColTasks = kRlSepTileCols * (kRlSepTileRows / kRlPix);
    static_assert(kColTasks <= NT && kRlSepTileCols == 32, "a 16-lane group of pass B reads sixteen different columns");
    const int tj = px % kRlSepTileCols, tq = (px / kRlSepTileCols) % (kRlSepTileRows / kRlPix);
    const int i0 = ti0 + kRlPix * tq, j = tj0 + tj;
    const bool col_task = px < kColTasks;
    const int idx0 = i0 * B.W + j;
    // Everything the block reads from memory is requested before anything is waited for — one round trip, not four:
    // the profiles (step 0 convolves with the PSF: reversed profiles in this upward walk; step 1 with its mirror
    // image), the update's other operand, which does not depend on the sums, and then the halo.
    float fy_v = 0.0f, fx_v = 0.0f, other[kRlPix] = {0.0f, 0.0f, 0.0f, 0.0f};
    // mode 1 (the reference's "same" convolution) walks the kernel downwards: step 0 takes the profiles reversed in this
    // upward walk, step 1 (the mirrored PSF) as they are.  Mode 0 (the reference's direct sums of kernels of <= 256 taps,
    // a[i + m - pr/2][j + n - pc/2] k[m][n]: a correlation) is the other way round — for the odd sizes every band PSF
    // has, correlating with k is convolving with its mirror image over the same halo.
    const bool rev = (step == 0) == (B.mode != 0);
    if (px < pc) fy_v = fy[rev ? pc - 1 - px : px];
    if (px < pr) fx_v = fx[rev ? pr - 1 - px : px];
    if (col_task && j < B.W) {
        const unsigned o_off = step == 0 ? B.off_d : B.off_u;
#pragma unroll
        for (int k = 0; k < kRlPix; ++k) other[k] = ws[i0 + k < B.H ? o_off + (unsigned)(idx0 + k * B.W) : B.off_zero];
    }
    {   // a wave per halo row, lanes along the row; zeros outside the image and beyond the halo's last column.
        // The tile's input was written by other CUs in the launch before, so every load is a trip to the far side
        // of the L2s: kRows rows x 2 column passes of loads are issued before the first is waited for (with the
        // plain loop the compiler kept four in flight, and the staging alone lasted eight round trips).
        // Rows are wave-uniform (scalar tests), the column tests of a pass are made once; a position outside the image
        // loads a stored zero, so the loads are unconditional, issue back to back and leave no masks to keep.
        const int wvu = THZ_UNIFORM(wv);
        auto stage = [&](int rb, int cb) {
            float v[kRows][2];
            bool cok[2];
            unsigned yv[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int c = cb + p * kWave, y = c0 + c;
                cok[p] = c < wsz && y >= 0 && y < B.W;
                yv[p] = (unsigned)y;
            }
#pragma unroll
            for (int k = 0; k < kRows; ++k) {
                const int r = rb + k * (NT / kWave), x = r0 + r;
                const bool row_ok = r < hs && x >= 0 && x < B.H;
                const unsigned row_at = a_off + (unsigned)(x * B.W);
#pragma unroll
                for (int p = 0; p < 2; ++p) v[k][p] = ws[row_ok && cok[p] ? row_at + yv[p] : B.off_zero];
            }
#pragma unroll
            for (int k = 0; k < kRows; ++k) {
                const int r = rb + k * (NT / kWave);
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int c = cb + p * kWave;
                    if (r < hs && c < wsp) a_s[r * wsp + c] = v[k][p];
                }
            }
        };
        // the first batch — the whole halo for kernels of up to 49 x 80 taps — stands in front of the loops: at a
        // loop head the compiler waits for every load in flight, which would put the profiles' round trip in
        // front of the halo's
        constexpr int kRowsPerBatch = kRows * (NT / kWave);
        stage(wvu, ln);
        for (int cb = ln + 2 * kWave; cb < wsp; cb += 2 * kWave) stage(wvu, cb);
        for (int rb = wvu + kRowsPerBatch; rb < hs; rb += kRowsPerBatch)
            for (int cb = ln; cb < wsp; cb += 2 * kWave) stage(rb, cb);
        if (px < nch * kRlChunk) fy_s[px] = fy_v;
        if (px < nchr * kRlChunk) fx_s[px] = fx_v;  // zeros behind the profile: whole chunks of taps
        for (int n = px + NT; n < nch * kRlChunk; n += NT)  // profiles of more than 1 024 taps
            fy_s[n] = n < pc ? fy[rev ? pc - 1 - n : n] : 0.0f;
        for (int m = px + NT; m < nchr * kRlChunk; m += NT) fx_s[m] = m < pr ? fx[rev ? pr - 1 - m : m] : 0.0f;
        // what a column window of pass B reaches behind the last halo row is multiplied by those zero taps: finite
        for (int n = px; n < kRlSepTileCols * (hsp - hs); n += NT) t_s[(n / (hsp - hs)) * hsp + hs + n % (hsp - hs)] = 0.0f;
    }
    __syncthreads();
    // pass A: (halo row, column group of four) per thread; T' takes the four sums as four 4-byte stores — a quarter
    // wave (16 consecutive halo rows, one column group) writes 16 consecutive floats of each of its four columns
    for (int task = px; task < (hs + 15) / 16 * (16 * kRlSepQuads); task += NT) {
        const int r = task / (16 * kRlSepQuads) * 16 + (task & 15), q = (task >> 4) % kRlSepQuads;  // a quarter wave: 16 rows, one column group
        if (r >= hs) continue;
        float acc[kRlPix];
        rl_tile_taps_split(a_s + r * wsp + 4 * q, wsp, fy_s, pc, 0, 1, acc);  // acc[p] = sum_n w[3 - p + n] fy_s[n]
        float *tc = t_s + (4 * q) * hsp + r;
        tc[0] = acc[3];
        tc[hsp] = acc[2];
        tc[2 * hsp] = acc[1];
        tc[3 * hsp] = acc[0];
    }
    __syncthreads();
    if (!col_task) return;
    // pass B: column tj, tile rows 4 tq .. 4 tq + 3: out[4 tq + k] = sum_m T'[tj][4 tq + k + m] fx_s[m] — the window
    // routine over a column of T' (round 3: it was a pixel per thread, one 4-byte read of T and one of fx per tap, the
    // longest stretch of a tile's LDS time; now 16-byte reads, a quarter of the threads, packed FMAs)
    float sum[kRlPix];
    rl_tile_taps_split(t_s + tj * hsp + kRlPix * tq, hsp, fx_s, pr, 0, 1, sum);  // sum[p] = sum_m w[3 - p + m] fx_s[m]
    if (j >= B.W) return;
#pragma unroll
    for (int k = 0; k < kRlPix; ++k) {
        if (i0 + k >= B.H) break;
        const float sm = sum[kRlPix - 1 - k];
        if (step == 0) ws[B.off_t + idx0 + k * B.W] = other[k] / (sm + 1e-12f);
        else ws[B.off_u + idx0 + k * B.W] = other[k] * sm;
    }
}

__global__ __launch_bounds__(256) void k_dc_filter_spectra(const float *__restrict__ filters, int n_bands,
                                                           int n_taps, const double *__restrict__ cs,
                                                           const double *__restrict__ sn, unsigned M,
                                                           unsigned nk, c32 *__restrict__ H)
{
#pragma clang fp contract(off)
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned k = e % nk;
    const int b = (int)(e / nk);
    if (b >= n_bands) return;
    const float *h = filters + (size_t)b * n_taps;
    double re = 0.0, im = 0.0;
    unsigned idx = 0;
    for (int j = 0; j < n_taps; ++j) {
        const double hj = (double)h[j];
        re += hj * cs[idx];
        im += hj * sn[idx];
        idx += k;
        if (idx >= M) idx -= M;
    }
    H[(size_t)b * nk + k] = c32{(float)(re / (double)M), (float)(im / (double)M)};
}

__global__ __launch_bounds__(256) void k_dc_gain(const RlBand *__restrict__ bands, int n_bands,
                                                 size_t npix, const float *__restrict__ energy,
                                                 const float *__restrict__ ws,
                                                 float *__restrict__ gain)
{
    const size_t total = (size_t)n_bands * npix;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx / npix);
        const size_t p = idx % npix;
        const RlBand B = bands[b];
        const int y = (int)(p / B.w), x = (int)(p % B.w);
        const float u = ws[B.off_u + (size_t)(y + B.pad_y) * B.W + (x + B.pad_x)];
        gain[idx] = sqrtf(fmaxf(u, 0.0f) / energy[idx]);
    }
}

// ---------------------------------------------------------------------------
// Synthetic cube generator (bench / test input): counter-based
// Philox4x32-10 so any tile is reproducible on host (tests/synth.py) or device.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t *out)
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    for (int r = 0; r < 10; ++r) {
        if (r > 0) { k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float u01_24(uint32_t r)
{
    return ((float)(r >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ float synth_pulse(float tt)
{
    const float z = tt / 0.35f;
    return -z * expf(-(z * z));
}

__device__ __forceinline__ float synth_sample(uint64_t trace, int i, int nt,
                                              const float *__restrict__ time, uint32_t seed,
                                              float A, float tc, float delta)
{
    const float tt = time[i];
    const float s = A * synth_pulse(tt - tc) + (0.3f * A) * synth_pulse(tt - tc - delta);
    const uint64_t g = trace * (uint64_t)nt + (uint64_t)i;
    const uint64_t blk = g >> 2;
    uint32_t w[4];
    philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), 0u, 0u, seed, 0u, w);
    const int l = (int)(g & 3);
    const float ua = u01_24(l < 2 ? w[0] : w[2]), ub = u01_24(l < 2 ? w[1] : w[3]);
    const float rad = sqrtf(-2.0f * logf(ua));
    const float ang = 6.28318548202514648438f * ub;
    const float nrm = (l & 1) ? rad * sinf(ang) : rad * cosf(ang);
    return s + (0.01f * A) * nrm;
}

__global__ __launch_bounds__(256) void k_synth(float *__restrict__ out, size_t ntraces, int nt,
                                               uint64_t first_trace,
                                               const float *__restrict__ time, uint32_t seed,
                                               int subtract_bias)
{
    const size_t total = ntraces * (size_t)nt;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const uint64_t trace = first_trace + idx / nt;
        const int i = (int)(idx % nt);
        uint32_t w[4];
        philox4x32_10((uint32_t)trace, (uint32_t)(trace >> 32), 0u, 1u, seed, 0u, w);
        const float A = 1.0f + 0.5f * u01_24(w[0]);
        const float tc = time[0] + 10.0f + 2.0f * u01_24(w[1]);
        const float delta = 3.0f + 17.0f * u01_24(w[2]);
        float v = synth_sample(trace, i, nt, time, seed, A, tc, delta);
        if (subtract_bias) v -= synth_sample(trace, 0, nt, time, seed, A, tc, delta);
        out[idx] = v;
    }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
static inline unsigned grid_1d(size_t total, unsigned block, unsigned cap)
{
    size_t g = (total + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

// Kernels that stage whole traces need more dynamic LDS than HIP's 64 KiB
// default; raise the per-kernel limit once per size.
template <class K>
static inline void allow_dynamic_lds(K kernel, size_t bytes)
{
#ifndef THZ_EMU
    static size_t allowed = 0;
    if (bytes > allowed) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        allowed = bytes;
    }
#else
    (void)kernel;
    (void)bytes;
#endif
}

static inline void wave_launch_geometry(const PlanDev &P, size_t npix, unsigned *grid,
                                        unsigned *block, size_t *lds)
{
    const unsigned wpb = (unsigned)P.waves_per_block;
    *block = wpb * kWave;
    *lds = (size_t)P.lds_per_wave * wpb;
    size_t blocks_per_cu = kLdsBytesPerCU / (*lds ? *lds : 1);
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    if (blocks_per_cu * wpb > 32) blocks_per_cu = 32 / wpb;
    size_t g = (npix + wpb - 1) / wpb;
    const size_t cap = (size_t)kNumCU * blocks_per_cu;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    *grid = (unsigned)g;
}

void launch_td_window(hipStream_t st, size_t npix, int nt, const float *in, const float *win,
                      float *out);

// F family: 8 waves per block share the twiddle tables in LDS; persistent grid.  A configuration
// whose tables do not leave room for eight wave buffers (nt = 4096 with a complex multiplier or with
// block accumulators) runs with as many waves as fit.
template <class PL, int CFG>
static unsigned f_block_threads()
{
    unsigned kBlock = 512;
#ifndef THZ_EMU
    if (const char *e = getenv("THZ_F_BLOCK")) {  // developer knob: waves per block
        const int v = atoi(e);
        if (v >= 64 && v <= 512 && v % 64 == 0) kBlock = (unsigned)v;
    }
#endif
    while (kBlock > kWave && PL::lds_bytes((int)(kBlock / kWave), CFG) > kLdsBytesPerCU) kBlock -= kWave;
    return kBlock;
}

// store-phase barriers of the F kernels (FArgs::bar); THZ_F_BAR overrides for A/B measurements
int g_f_bar_override = -1;  // tests (the emulation harness) select a barrier mode here
static int f_bar_mode()
{
    if (g_f_bar_override >= 0) return g_f_bar_override & 7;
#ifndef THZ_EMU
    if (const char *e = getenv("THZ_F_BAR")) return atoi(e) & 7;
#endif
    return kFBarDefault;
}

int g_grid_cap_override = 0;  // tests (the emulation harness): at most this many blocks for the F / P transform kernels,
                              // so that a handful of traces already takes a block through several rounds
template <class PL, int MODE, int CFG>
static size_t f_grid(size_t npix)
{
    const unsigned kWpb = f_block_threads<PL, CFG>() / kWave;
    const size_t lds = PL::lds_bytes((int)kWpb, CFG);
    size_t per_cu = kLdsBytesPerCU / lds;
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 2) per_cu = 2;
    size_t g = (npix + kWpb - 1) / kWpb;
    if (g > (size_t)kNumCU * per_cu) g = (size_t)kNumCU * per_cu;
    if (g_grid_cap_override > 0 && g > (size_t)g_grid_cap_override) g = (size_t)g_grid_cap_override;
    if (g < 1) g = 1;
    return g;
}

template <class PL, int MODE, int CFG>
static void launch_f(hipStream_t st, const PlanDev &P, const FArgs &A)
{
    const unsigned kBlock = f_block_threads<PL, CFG>();
    const size_t lds = PL::lds_bytes((int)(kBlock / kWave), CFG);
    const size_t g = f_grid<PL, MODE, CFG>(A.npix);
    FTables T{reinterpret_cast<const cx *>(P.f_t1), reinterpret_cast<const cx *>(P.f_t2),
              reinterpret_cast<const cx *>(P.f_w2n)};
    allow_dynamic_lds(k_f<PL, MODE, CFG>, lds);
    FArgs B = A;
    B.bar = (CFG & kCfgBar) ? f_bar_mode() : 0;
    THZ_LAUNCH((k_f<PL, MODE, CFG>), (unsigned)g, kBlock, lds, st, B, T);
}

template <int MODE, int CFG>
static void dispatch_f_size(hipStream_t st, const PlanDev &P, const FArgs &A)
{
    switch (P.nt) {
    case 4096: launch_f<FPlan4096, MODE, CFG>(st, P, A); break;
    case 2048: launch_f<FPlan2048, MODE, CFG>(st, P, A); break;
    default: launch_f<FPlan1024, MODE, CFG>(st, P, A); break;
    }
}

// the band-limited complex multiplier table (kCfgBand) is built for the one configuration it buys a wave for: nt = 4096
// with the in-launch sums; the caller's band must fit the table
static bool f_band_fits(const PlanDev &P, int lo4, int n)
{
    return P.nt == 4096 && n > 0 && n <= FPlan4096::BAND_BINS && lo4 >= 0 && lo4 % 4 == 0 && n % 4 == 0;
}

template <int MODE>
static void dispatch_f(hipStream_t st, const PlanDev &P, const FArgs &A, bool amp_phase)
{
    const int bar = f_bar_mode() ? kCfgBar : 0;
    if constexpr (MODE == kInv) {
        if (bar) dispatch_f_size<MODE, kCfgBar>(st, P, A);
        else dispatch_f_size<MODE, 0>(st, P, A);
    } else {
        // the fused chain always writes amplitudes and phases
        int cfg = ((MODE == kPipe || amp_phase) ? kCfgAmpPhase : 0) | (A.cmask ? kCfgCMask : 0) | bar;
        if constexpr (MODE == kPipe) {
            if (A.sum_partial) {  // pixel sums inside the launch: the block-uniform trace loop of the barrier builds
                if (A.cmask && f_band_fits(P, A.band_lo4, A.band_n))   // nt = 4096: the band-limited table keeps the eighth wave
                    launch_f<FPlan4096, MODE, kCfgBar | kCfgAmpPhase | kCfgCMask | kCfgSums | kCfgBand>(st, P, A);
                else if (A.cmask) dispatch_f_size<MODE, kCfgBar | kCfgAmpPhase | kCfgCMask | kCfgSums>(st, P, A);
                else dispatch_f_size<MODE, kCfgBar | kCfgAmpPhase | kCfgSums>(st, P, A);
                return;
            }
        }
        switch (cfg) {
        case kCfgBar | kCfgAmpPhase | kCfgCMask: dispatch_f_size<MODE, kCfgBar | kCfgAmpPhase | kCfgCMask>(st, P, A); break;
        case kCfgBar | kCfgAmpPhase: dispatch_f_size<MODE, kCfgBar | kCfgAmpPhase>(st, P, A); break;
        case kCfgAmpPhase | kCfgCMask: dispatch_f_size<MODE, kCfgAmpPhase | kCfgCMask>(st, P, A); break;
        case kCfgAmpPhase: dispatch_f_size<MODE, kCfgAmpPhase>(st, P, A); break;
        default:
            if constexpr (MODE == kFwd) {
                switch (cfg) {
                case kCfgBar | kCfgCMask: dispatch_f_size<MODE, kCfgBar | kCfgCMask>(st, P, A); break;
                case kCfgBar: dispatch_f_size<MODE, kCfgBar>(st, P, A); break;
                case kCfgCMask: dispatch_f_size<MODE, kCfgCMask>(st, P, A); break;
                default: dispatch_f_size<MODE, 0>(st, P, A); break;
                }
            }
            break;
        }
    }
}

template <class PL, int MODE>
static void launch_fb(hipStream_t st, const PlanDev &P, const FBArgs &A)
{
    const unsigned kBlock = 512, kWpb = kBlock / kWave;
    const size_t lds = FBLayout<PL>::lds_bytes((int)kWpb, P.nt, P.nf);
    size_t per_cu = kLdsBytesPerCU / lds;
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 2) per_cu = 2;
    const size_t n_pairs = (A.npix + 1) / 2;  // one wave per pair of traces
    size_t g = (n_pairs + kWpb - 1) / kWpb;
    if (g > (size_t)kNumCU * per_cu) g = (size_t)kNumCU * per_cu;
    if (g < 1) g = 1;
    FTables T{reinterpret_cast<const cx *>(P.f_t1), reinterpret_cast<const cx *>(P.f_t2), nullptr};
    allow_dynamic_lds(k_fb<PL, MODE>, lds);
    THZ_LAUNCH((k_fb<PL, MODE>), (unsigned)g, kBlock, lds, st, A, T);
}

template <int MODE>
static void dispatch_fb(hipStream_t st, const PlanDev &P, FBArgs &A)
{
    A.nt = P.nt;
    A.nf = P.nf;
    A.w = reinterpret_cast<const cx *>(P.chirp_conj);
    A.bf = reinterpret_cast<const cx *>(P.bfft);
    switch (1 << P.log2n) {
    case 2048: launch_fb<FPlan4096, MODE>(st, P, A); break;
    case 1024: launch_fb<FPlan2048, MODE>(st, P, A); break;
    default: launch_fb<FPlan1024, MODE>(st, P, A); break;
    }
}

// P family (fft_p.hpp): Q pairs of traces per wave (kPPairs; THZ_P_PAIRS overrides for A/B measurements), up to
// 16 / Q waves per block share the tables
int g_p_pairs_override = 0;  // tests (the emulation harness) select Q here
static int p_pairs()
{
    if (g_p_pairs_override == 1 || g_p_pairs_override == 2) return g_p_pairs_override;
#ifndef THZ_EMU
    if (const char *e = getenv("THZ_P_PAIRS")) {
        const int v = atoi(e);
        if (v == 1 || v == 2) return v;
    }
#endif
    return kPPairsDefault;
}

template <class PL, int Q, bool CM, bool SUMS>
static void p_geometry(size_t npix, unsigned *waves_out, size_t *grid_out, size_t *lds_out)
{
    const unsigned waves = (unsigned)p_block_waves<PL>(Q, CM, SUMS);
    const size_t n_units = (npix + 2 * Q - 1) / (2 * Q);
    size_t g = (n_units + waves - 1) / waves;
    if (g > (size_t)kNumCU) g = kNumCU;
    if (g_grid_cap_override > 0 && g > (size_t)g_grid_cap_override) g = (size_t)g_grid_cap_override;
    if (g < 1) g = 1;
    *waves_out = waves;
    *grid_out = g;
    *lds_out = PL::lds_bytes((int)waves, Q, CM, SUMS);
}

template <class PL, int MODE, int Q, bool CM, bool SUMS = false>
static void launch_p(hipStream_t st, const PlanDev &P, const FBArgs &A)
{
    unsigned waves;
    size_t g, lds;
    p_geometry<PL, Q, CM, SUMS>(A.npix, &waves, &g, &lds);
    PTables T{reinterpret_cast<const cx *>(P.p_t1), reinterpret_cast<const cx *>(P.p_t2)};
    allow_dynamic_lds(k_p<PL, MODE, Q, CM, SUMS>, lds);
    THZ_LAUNCH((k_p<PL, MODE, Q, CM, SUMS>), (unsigned)g, waves * kWave, lds, st, A, T);
}

template <class PL, int MODE, bool ALLOW_TWO = true>
static void launch_p_variant(hipStream_t st, const PlanDev &P, const FBArgs &A)
{
    const bool two = ALLOW_TWO && p_pairs() == 2;
    if constexpr (MODE == kPipe) {
        if (A.sum_partial && !two) {  // pixel sums inside the launch (PSums)
            if (A.cmask) launch_p<PL, MODE, 1, true, true>(st, P, A);
            else launch_p<PL, MODE, 1, false, true>(st, P, A);
            return;
        }
    }
    if constexpr (MODE != kInv) {
        if (A.cmask) {  // complex multiplier on top of the band pass
            if constexpr (ALLOW_TWO)
                if (two) { launch_p<PL, MODE, 2, true>(st, P, A); return; }
            launch_p<PL, MODE, 1, true>(st, P, A);
            return;
        }
    }
    if constexpr (ALLOW_TWO)
        if (two) { launch_p<PL, MODE, 2, false>(st, P, A); return; }
    launch_p<PL, MODE, 1, false>(st, P, A);
}

// rows of the partial-sum workspace of a P launch with in-launch sums (one per block), 0 when it has none
template <class PL, bool ALLOW_TWO = true>
static size_t p_sum_rows(size_t npix, bool cmask)
{
    if (ALLOW_TWO && p_pairs() == 2) return 0;
    unsigned waves;
    size_t g, lds;
    if (cmask) p_geometry<PL, 1, true, true>(npix, &waves, &g, &lds);
    else p_geometry<PL, 1, false, true>(npix, &waves, &g, &lds);
    return g;
}

template <int MODE>
static void dispatch_p(hipStream_t st, const PlanDev &P, FBArgs &A)
{
    A.nt = P.nt;
    A.nf = P.nf;
    switch (P.nt) {
    case 1001: launch_p_variant<PPlan1001, MODE>(st, P, A); break;
    case 1200: launch_p_variant<PPlan1200, MODE, false>(st, P, A); break;
    case 1500: launch_p_variant<PPlan1500, MODE, false>(st, P, A); break;
    case 2000: launch_p_variant<PPlan2000, MODE, false>(st, P, A); break;
    default: launch_p_variant<PPlan1000, MODE>(st, P, A); break;
    }
}

// PH kernels (fft_ph.hpp): even lengths whose half is a P plan
template <class PL, int MODE>
static void launch_ph(hipStream_t st, const PlanDev &P, FBArgs &A)
{
    A.nt = P.nt;
    A.nf = P.nf;
    const unsigned waves = (unsigned)PHLayout<PL>::waves();
    size_t g = (A.npix + waves - 1) / waves;
    if (g > (size_t)kNumCU) g = kNumCU;
    if (g_grid_cap_override > 0 && g > (size_t)g_grid_cap_override) g = (size_t)g_grid_cap_override;
    if (g < 1) g = 1;
    const size_t lds = PHLayout<PL>::lds_bytes((int)waves);
    PHTables T{reinterpret_cast<const cx *>(P.p_t1), reinterpret_cast<const cx *>(P.p_t2),
               reinterpret_cast<const cx *>(P.p_t2) + PL::T2_ENTRIES};
    allow_dynamic_lds(k_ph<PL, MODE>, lds);
    THZ_LAUNCH((k_ph<PL, MODE>), (unsigned)g, waves * kWave, lds, st, A, T);
}

template <int MODE>
static void dispatch_ph(hipStream_t st, const PlanDev &P, FBArgs &A)
{
    switch (P.half_n) {
    case 1001: launch_ph<PPlan1001, MODE>(st, P, A); break;
    case 1200: launch_ph<PPlan1200, MODE>(st, P, A); break;
    case 1500: launch_ph<PPlan1500, MODE>(st, P, A); break;
    case 2000: launch_ph<PPlan2000, MODE>(st, P, A); break;
    default: launch_ph<PPlan1000, MODE>(st, P, A); break;
    }
}

// FBC: S waves per pair (fft_fb.hpp), forward and inverse as separate kernels, for the lengths
// 1024 < nt < 8192 that are not a power of two

template <int S, int MODE>
static void launch_fbc(hipStream_t st, const PlanDev &P, FB2Args &B)
{
    FBArgs &A = B.a;
    A.nt = P.nt;
    A.nf = P.nf;
    A.w = reinterpret_cast<const cx *>(P.chirp_conj);
    A.bf = reinterpret_cast<const cx *>(P.bfft);
    B.tw = reinterpret_cast<const cx *>(P.tw);
    using PL = FPlan4096;
    using LY = FBCLayout<PL, S>;
    const size_t lds = LY::lds_bytes();
    const size_t n_pairs = (A.npix + 1) / 2;
    size_t g = (n_pairs + LY::kPairs - 1) / LY::kPairs;
    if (g > (size_t)kNumCU) g = kNumCU;
    if (g < 1) g = 1;
    FTables T{reinterpret_cast<const cx *>(P.f_t1), reinterpret_cast<const cx *>(P.f_t2), nullptr};
    allow_dynamic_lds(k_fbc<PL, S, MODE>, lds);
    THZ_LAUNCH((k_fbc<PL, S, MODE>), (unsigned)g, 512, lds, st, B, T);
}

template <int MODE>
static void dispatch_fbc(hipStream_t st, const PlanDev &P, FB2Args &B)
{
    switch (P.family) {
    case kFamilyFB2: launch_fbc<2, MODE>(st, P, B); break;
    case kFamilyFB4: launch_fbc<4, MODE>(st, P, B); break;
    default: launch_fbc<8, MODE>(st, P, B); break;
    }
}

void launch_fft_fwd(hipStream_t st, const PlanDev &P, size_t npix, const float *in,
                    const float *wa, const float *wb, float *data_out, c32 *fft_out,
                    float *amp_out, float *ph_out, const float *mask, const c32 *cmask)
{
    // A complex multiplier is fused by the F and P kernels; every other family multiplies in a second
    // (elementwise) launch over the stored spectrum and amplitudes.
    const bool p_direct = P.family == kFamilyP && !wb && !data_out;
    if (cmask && !p_direct && !(P.family == kFamilyF && fft_out && ((amp_out != nullptr) == (ph_out != nullptr)))) {
        launch_fft_fwd(st, P, npix, in, wa, wb, data_out, fft_out, amp_out, ph_out, mask, nullptr);
        launch_fd_cmask(st, npix, P.nf, P.nt, fft_out, amp_out, cmask);
        return;
    }
    // F kernels: one window, spectrum required, |X| and phase both or neither,
    // no windowed-trace output.  When the stage's `data` output is wanted the
    // multiply runs as its own elementwise launch first (bit-identical: the
    // same single f32 multiply per window).
    if (P.family == kFamilyF && data_out && wa && fft_out && ((amp_out != nullptr) == (ph_out != nullptr))) {
        launch_td_window(st, npix, P.nt, in, wa, data_out);
        if (wb) launch_td_window(st, npix, P.nt, data_out, wb, data_out);
        launch_fft_fwd(st, P, npix, data_out, nullptr, nullptr, nullptr, fft_out, amp_out, ph_out, mask, cmask);
        return;
    }
    // FB kernels (chirp-z lengths): same split — the windowed-trace output is its own launch
    if (P.family == kFamilyFB2 || P.family == kFamilyFB4 || P.family == kFamilyFB8) {
        if (data_out && wa) {
            launch_td_window(st, npix, P.nt, in, wa, data_out);
            if (wb) launch_td_window(st, npix, P.nt, data_out, wb, data_out);
            launch_fft_fwd(st, P, npix, data_out, nullptr, nullptr, nullptr, fft_out, amp_out, ph_out, mask);
            return;
        }
        if (!wb && !data_out && P.half_n) {  // half-length mixed-radix transform + split (fft_ph.hpp)
            FBArgs A{};
            A.npix = npix; A.in = in; A.pre_win = wa; A.mask = mask ? mask : P.ones;
            A.fft_out = reinterpret_cast<cx *>(fft_out); A.amp_out = amp_out; A.ph_out = ph_out;
            dispatch_ph<kFwd>(st, P, A);
            return;
        }
        if (!wb && !data_out) {
            FB2Args B{};
            B.a.npix = npix; B.a.in = in; B.a.pre_win = wa; B.a.mask = mask ? mask : P.ones;
            B.a.fft_out = reinterpret_cast<cx *>(fft_out); B.a.amp_out = amp_out; B.a.ph_out = ph_out;
            dispatch_fbc<kFwd>(st, P, B);
            return;
        }
    }
    if (P.family == kFamilyP && !wb && !data_out) {
        FBArgs A{};
        A.npix = npix; A.in = in; A.pre_win = wa; A.mask = mask ? mask : P.ones;
        A.fft_out = reinterpret_cast<cx *>(fft_out); A.amp_out = amp_out; A.ph_out = ph_out;
        A.cmask = reinterpret_cast<const cx *>(cmask);
        dispatch_p<kFwd>(st, P, A);
        return;
    }
    if ((P.family == kFamilyFB || P.family == kFamilyP) && data_out && wa) {
        launch_td_window(st, npix, P.nt, in, wa, data_out);
        if (wb) launch_td_window(st, npix, P.nt, data_out, wb, data_out);
        launch_fft_fwd(st, P, npix, data_out, nullptr, nullptr, nullptr, fft_out, amp_out, ph_out, mask, cmask);
        return;
    }
    if (P.family == kFamilyFB && !wb && !data_out) {
        FBArgs A{};
        A.npix = npix; A.in = in; A.pre_win = wa; A.mask = mask ? mask : P.ones;
        A.fft_out = reinterpret_cast<cx *>(fft_out); A.amp_out = amp_out; A.ph_out = ph_out;
        dispatch_fb<kFwd>(st, P, A);
        return;
    }
    if (P.family == kFamilyF && !wb && !data_out && fft_out && ((amp_out != nullptr) == (ph_out != nullptr))) {
        FArgs A{};
        A.npix = npix; A.in = in; A.pre_win = wa; A.fft_out = reinterpret_cast<cx *>(fft_out); A.amp_out = amp_out;
        A.ph_out = ph_out; A.mask = mask ? mask : P.ones; A.cmask = reinterpret_cast<const cx *>(cmask);
        dispatch_f<kFwd>(st, P, A, amp_out != nullptr);
        return;
    }
    if (P.big_scratch) {  // buffers in global scratch: four waves per block, at most big_waves waves
        unsigned g = (unsigned)((npix + 3) / 4);
        if (g > (unsigned)(P.big_waves / 4)) g = (unsigned)(P.big_waves / 4);
        if (g_grid_cap_override > 0 && g > (unsigned)g_grid_cap_override) g = (unsigned)g_grid_cap_override;
        THZ_LAUNCH(k_fft_fwd_big, g ? g : 1, 256, 0, st, P, npix, in, wa, wb, data_out, fft_out, amp_out, ph_out, mask);
        return;
    }
    unsigned grid, block;
    size_t lds;
    wave_launch_geometry(P, npix, &grid, &block, &lds);
    allow_dynamic_lds(k_fft_fwd, lds);
    THZ_LAUNCH(k_fft_fwd, grid, block, lds, st, P, npix, in, wa, wb, data_out, fft_out, amp_out,
               ph_out, mask);
}

void launch_fft_inv(hipStream_t st, const PlanDev &P, size_t npix, const c32 *fft_in,
                    const float *win, float *out, float *img)
{
    if (P.family == kFamilyF) {
        FArgs A{};
        A.npix = npix; A.fft_in = reinterpret_cast<const cx *>(fft_in); A.post_win = win; A.data_out = out; A.img = img;
        dispatch_f<kInv>(st, P, A, false);
        return;
    }
    if (P.family == kFamilyP) {
        FBArgs A{};
        A.npix = npix; A.fft_in = reinterpret_cast<const cx *>(fft_in); A.mask = P.ones; A.post_win = win;
        A.data_out = out; A.img = img;
        dispatch_p<kInv>(st, P, A);
        return;
    }
    if (P.family == kFamilyFB) {
        FBArgs A{};
        A.npix = npix; A.fft_in = reinterpret_cast<const cx *>(fft_in); A.mask = P.ones; A.post_win = win;
        A.data_out = out; A.img = img;
        dispatch_fb<kInv>(st, P, A);
        return;
    }
    if ((P.family == kFamilyFB2 || P.family == kFamilyFB4 || P.family == kFamilyFB8) && P.half_n) {
        FBArgs A{};
        A.npix = npix; A.fft_in = reinterpret_cast<const cx *>(fft_in); A.mask = P.ones; A.post_win = win;
        A.data_out = out; A.img = img;
        dispatch_ph<kInv>(st, P, A);
        return;
    }
    if (P.family == kFamilyFB2 || P.family == kFamilyFB4 || P.family == kFamilyFB8) {
        FB2Args B{};
        B.a.npix = npix; B.a.fft_in = reinterpret_cast<const cx *>(fft_in); B.a.mask = P.ones; B.a.post_win = win;
        B.a.data_out = out; B.a.img = img;
        dispatch_fbc<kInv>(st, P, B);
        return;
    }
    if (P.big_scratch) {
        unsigned g = (unsigned)((npix + 3) / 4);
        if (g > (unsigned)(P.big_waves / 4)) g = (unsigned)(P.big_waves / 4);
        if (g_grid_cap_override > 0 && g > (unsigned)g_grid_cap_override) g = (unsigned)g_grid_cap_override;
        THZ_LAUNCH(k_fft_inv_big, g ? g : 1, 256, 0, st, P, npix, fft_in, win, out, img);
        return;
    }
    unsigned grid, block;
    size_t lds;
    wave_launch_geometry(P, npix, &grid, &block, &lds);
    allow_dynamic_lds(k_fft_inv, lds);
    THZ_LAUNCH(k_fft_inv, grid, block, lds, st, P, npix, fft_in, win, out, img);
}

// Rows of the partial-sum workspace (2 nf floats each) a fused launch with in-kernel pixel sums needs for npix
// traces — one per block of its grid — or 0 when this plan has no fused kernel that sums (other families).
template <class PL>
static size_t f_sum_rows(size_t npix, bool cmask, bool band = false)
{
    if (cmask && band) {
        constexpr int CFG = kCfgBar | kCfgAmpPhase | kCfgCMask | kCfgSums | kCfgBand;
        return f_grid<PL, kPipe, CFG>(npix);
    }
    if (cmask) {
        constexpr int CFG = kCfgBar | kCfgAmpPhase | kCfgCMask | kCfgSums;
        return f_grid<PL, kPipe, CFG>(npix);
    }
    constexpr int CFG = kCfgBar | kCfgAmpPhase | kCfgSums;
    return f_grid<PL, kPipe, CFG>(npix);
}

size_t pipeline_sum_rows(const PlanDev &P, size_t npix, bool cmask, int band_lo4, int band_n)
{
    if (npix == 0) return 0;
    if (P.family == kFamilyP) {
        switch (P.nt) {
        case 1001: return p_sum_rows<PPlan1001>(npix, cmask);
        case 1200: return p_sum_rows<PPlan1200, false>(npix, cmask);
        case 1500: return p_sum_rows<PPlan1500, false>(npix, cmask);
        case 2000: return p_sum_rows<PPlan2000, false>(npix, cmask);
        default: return p_sum_rows<PPlan1000>(npix, cmask);
        }
    }
    if (P.family != kFamilyF) return 0;
    switch (P.nt) {
    case 4096: return f_sum_rows<FPlan4096>(npix, cmask, cmask && f_band_fits(P, band_lo4, band_n));
    case 2048: return f_sum_rows<FPlan2048>(npix, cmask);
    default: return f_sum_rows<FPlan1024>(npix, cmask);
    }
}

void launch_pipeline(hipStream_t st, const PlanDev &P, size_t npix, const float *raw,
                     const float *pre_win, const float *mask, const float *post_win, c32 *fft_out,
                     float *amp_out, float *ph_out, float *data_out, float *img, const c32 *cmask,
                     float *sum_partial, int band_lo4, int band_n)
{
    if (P.family == kFamilyF && fft_out && amp_out && ph_out) {
        FArgs A{};
        A.npix = npix; A.in = raw; A.pre_win = pre_win; A.fft_out = reinterpret_cast<cx *>(fft_out); A.amp_out = amp_out;
        A.ph_out = ph_out; A.mask = mask ? mask : P.ones; A.post_win = post_win;
        A.cmask = reinterpret_cast<const cx *>(cmask);
        A.data_out = data_out; A.img = img;
        A.sum_partial = sum_partial;  // pipeline_sum_rows(P, npix, cmask) rows, or null
        A.band_lo4 = band_lo4; A.band_n = band_n;
        dispatch_f<kPipe>(st, P, A, true);
        return;
    }
    if (cmask && !(P.family == kFamilyP && fft_out && amp_out && ph_out && data_out)) {
        // every other family: the complex multiply is its own pass over the stored spectrum
        launch_fft_fwd(st, P, npix, raw, pre_win, nullptr, nullptr, fft_out, amp_out, ph_out, mask, cmask);
        launch_fft_inv(st, P, npix, fft_out, post_win, data_out, img);
        return;
    }
    if ((P.family == kFamilyFB2 || P.family == kFamilyFB4 || P.family == kFamilyFB8) && P.half_n && fft_out && amp_out && ph_out && data_out) {
        FBArgs A{};  // one launch: half-length transform, split, epilogue, merge, inverse (fft_ph.hpp)
        A.npix = npix; A.in = raw; A.pre_win = pre_win; A.mask = mask ? mask : P.ones;
        A.post_win = post_win; A.fft_out = reinterpret_cast<cx *>(fft_out); A.amp_out = amp_out; A.ph_out = ph_out;
        A.data_out = data_out; A.img = img;
        dispatch_ph<kPipe>(st, P, A);
        return;
    }
    if ((P.family == kFamilyFB2 || P.family == kFamilyFB4 || P.family == kFamilyFB8) && fft_out && data_out) {
        // S regions per pair leave no room to carry the pair's spectra from the forward to the inverse
        // transform inside one kernel: two launches, the inverse reads the masked spectra back (the
        // round trip through HBM is noise next to 4 S core runs per pair)
        launch_fft_fwd(st, P, npix, raw, pre_win, nullptr, nullptr, fft_out, amp_out, ph_out, mask);
        launch_fft_inv(st, P, npix, fft_out, post_win, data_out, img);
        return;
    }
    if (P.family == kFamilyP && fft_out && amp_out && ph_out && data_out) {
        FBArgs A{};
        A.npix = npix; A.in = raw; A.pre_win = pre_win; A.mask = mask ? mask : P.ones;
        A.post_win = post_win; A.fft_out = reinterpret_cast<cx *>(fft_out); A.amp_out = amp_out; A.ph_out = ph_out;
        A.data_out = data_out; A.img = img;
        A.cmask = reinterpret_cast<const cx *>(cmask);
        A.sum_partial = sum_partial;
        dispatch_p<kPipe>(st, P, A);
        return;
    }
    if (P.family == kFamilyFB && fft_out && amp_out && ph_out && data_out) {
        FBArgs A{};
        A.npix = npix; A.in = raw; A.pre_win = pre_win; A.mask = mask ? mask : P.ones;
        A.post_win = post_win; A.fft_out = reinterpret_cast<cx *>(fft_out); A.amp_out = amp_out; A.ph_out = ph_out;
        A.data_out = data_out; A.img = img;
        dispatch_fb<kPipe>(st, P, A);
        return;
    }
    if (P.big_scratch && fft_out) {  // long traces: a forward and an inverse launch around the stored spectrum
        launch_fft_fwd(st, P, npix, raw, pre_win, nullptr, nullptr, fft_out, amp_out, ph_out, mask);
        launch_fft_inv(st, P, npix, fft_out, post_win, data_out, img);
        return;
    }
    if (P.mode != kModePow2 && fft_out) {
        // generic chirp-z plan (thz_set_kernel_family(1) with a length that is not a power of two): k_pipeline below
        // is the power-of-two chain only, so the chain is a forward and an inverse launch around the stored spectrum
        launch_fft_fwd(st, P, npix, raw, pre_win, nullptr, nullptr, fft_out, amp_out, ph_out, mask);
        launch_fft_inv(st, P, npix, fft_out, post_win, data_out, img);
        return;
    }
    unsigned grid, block;
    size_t lds;
    wave_launch_geometry(P, npix, &grid, &block, &lds);
    allow_dynamic_lds(k_pipeline, lds);
    THZ_LAUNCH(k_pipeline, grid, block, lds, st, P, npix, raw, pre_win, mask, post_win, fft_out,
               amp_out, ph_out, data_out, img);
}

void launch_fd_mask(hipStream_t st, size_t npix, int nf, c32 *fft, float *amp, const float *mask)
{
    THZ_LAUNCH(k_fd_mask, grid_1d(npix * nf, 256, kNumCU * 8), 256, 0, st, npix, nf, fft, amp,
               mask);
}

void launch_fd_cmask(hipStream_t st, size_t npix, int nf, int nt, c32 *fft, float *amp,
                     const c32 *cmask)
{
    THZ_LAUNCH(k_fd_cmask, grid_1d(npix * nf, 256, kNumCU * 8), 256, 0, st, npix, nf, nt, fft, amp,
               cmask);
}

void launch_scale_vec(hipStream_t st, const float *in, float f, size_t n, float *out)
{
    THZ_LAUNCH(k_scale_vec, grid_1d(n, 256, kNumCU * 8), 256, 0, st, in, f, n, out);
}
void launch_add_vec(hipStream_t st, float *dst, const float *src, size_t n)
{
    THZ_LAUNCH(k_add_vec, grid_1d(n, 256, kNumCU * 8), 256, 0, st, dst, src, n);
}
void launch_add_u64(hipStream_t st, unsigned long long *dst, const unsigned long long *src, size_t n)
{
    THZ_LAUNCH(k_add_u64, grid_1d(n, 256, kNumCU * 8), 256, 0, st, dst, src, n);
}

void launch_td_window(hipStream_t st, size_t npix, int nt, const float *in, const float *win,
                      float *out)
{
    const bool vec = nt % 4 == 0 && ((uintptr_t)in | (uintptr_t)win | (uintptr_t)out) % 16 == 0;
    const unsigned grid = grid_1d(npix * kWave, 256, kNumCU * 8);
    if (vec && nt % 256 == 0 && nt <= 4096) {  // the window chunks of a lane in registers
        switch (nt / 256) {
        case 1: THZ_LAUNCH(k_td_window_regs<1>, grid, 256, 0, st, npix, in, win, out); return;
        case 2: THZ_LAUNCH(k_td_window_regs<2>, grid, 256, 0, st, npix, in, win, out); return;
        case 4: THZ_LAUNCH(k_td_window_regs<4>, grid, 256, 0, st, npix, in, win, out); return;
        case 8: THZ_LAUNCH(k_td_window_regs<8>, grid, 256, 0, st, npix, in, win, out); return;
        case 16: THZ_LAUNCH(k_td_window_regs<16>, grid, 256, 0, st, npix, in, win, out); return;
        default: break;
        }
    }
    if (vec) THZ_LAUNCH(k_td_window<true>, grid, 256, 0, st, npix, nt, in, win, out);
    else THZ_LAUNCH(k_td_window<false>, grid, 256, 0, st, npix, nt, in, win, out);
}

void launch_intensity(hipStream_t st, size_t npix, int nt, float *data, float *img,
                      int subtract_bias)
{
    const unsigned grid = grid_1d(npix * kWave, 256, kNumCU * 8);
    if (nt % 4 == 0 && (uintptr_t)data % 16 == 0)
        THZ_LAUNCH(k_intensity<true>, grid, 256, 0, st, npix, nt, data, img, subtract_bias);
    else
        THZ_LAUNCH(k_intensity<false>, grid, 256, 0, st, npix, nt, data, img, subtract_bias);
}

void launch_sum_axis0(hipStream_t st, const float *arr, size_t n0, size_t inner, float div,
                      float *out, const float *carry)
{
    THZ_LAUNCH(k_sum_axis0, grid_1d(inner, 256, kNumCU * 16), 256, 0, st, arr, n0, inner, div, out, carry);
}

void launch_sum_rows_f64(hipStream_t st, const float *arr, size_t n0, size_t inner, float *out)
{
    THZ_LAUNCH(k_sum_rows_f64, grid_1d(inner, 256, kNumCU * 16), 256, 0, st, arr, n0, inner, out);
}

// returns the number of partial rows written to `partial` (each L floats); 0 if L is
// too wide for the kernel (caller falls back)
size_t launch_colsum_partial(hipStream_t st, const float *arr, size_t nrows, size_t L,
                             float *partial, size_t max_groups, const uint32_t *list)
{
    const size_t chunks = (L / 4 + 255) / 256;  // 16-byte chunks per thread
    if (chunks > 8) return 0;
    size_t groups = (size_t)kNumCU * 8;
    if (groups > max_groups) groups = max_groups;
    if (groups > nrows) groups = nrows;
    if (groups < 1) groups = 1;
    const size_t rows_per_group = (nrows + groups - 1) / groups;
    groups = (nrows + rows_per_group - 1) / rows_per_group;
    if (chunks <= 1) THZ_LAUNCH((k_colsum_partial<1>), (unsigned)groups, 256, 0, st, arr, nrows, L, rows_per_group, partial, list);
    else if (chunks <= 2) THZ_LAUNCH((k_colsum_partial<2>), (unsigned)groups, 256, 0, st, arr, nrows, L, rows_per_group, partial, list);
    else if (chunks <= 3) THZ_LAUNCH((k_colsum_partial<3>), (unsigned)groups, 256, 0, st, arr, nrows, L, rows_per_group, partial, list);
    else if (chunks <= 5) THZ_LAUNCH((k_colsum_partial<5>), (unsigned)groups, 256, 0, st, arr, nrows, L, rows_per_group, partial, list);
    else THZ_LAUNCH((k_colsum_partial<8>), (unsigned)groups, 256, 0, st, arr, nrows, L, rows_per_group, partial, list);
    return groups;
}

void launch_roi_mask(hipStream_t st, const uint64_t *d_poly, int n, uint64_t x_min, uint64_t x_max,
                     uint64_t y_min, uint64_t y_max, uint64_t x_size, uint64_t y_size,
                     uint8_t *d_mask)
{
    THZ_LAUNCH(k_roi_mask, grid_1d(x_size * y_size, 256, kNumCU * 8), 256, 0, st, d_poly, n, x_min,
               x_max, y_min, y_max, x_size, y_size, d_mask);
}

void launch_gather_sum(hipStream_t st, const float *arr, size_t len, const uint32_t *d_list,
                       uint32_t count, float div, float *out)
{
    THZ_LAUNCH(k_gather_sum, (unsigned)((len + 63) / 64), 64, 0, st, arr, len, d_list, count, div, out);
}

void launch_gather_sum_w(hipStream_t st, const float *arr, size_t len, const uint32_t *d_list, uint32_t count, float div,
                         const float *w1, const float *w2, const float *w3, float *out)
{
    THZ_LAUNCH(k_gather_sum_w, (unsigned)((len + 63) / 64), 64, 0, st, arr, len, d_list, count, div, w1, w2, w3, out);
}

void launch_scale_rows_partial(hipStream_t st, const float *arr, size_t m, size_t ny, size_t L, size_t s, const float *carry, float div,
                               float *out)
{
    THZ_LAUNCH(k_scale_rows_partial, grid_1d((ny / s) * L, 256, kNumCU * 8), 256, 0, st, arr, m, ny, L, s, carry, div, out);
}

void launch_div_vec(hipStream_t st, const float *in, const float *w, float d, size_t n, float *out)
{
    THZ_LAUNCH(k_div_vec, grid_1d(n, 256, kNumCU), 256, 0, st, in, w, d, n, out);
}

void launch_scale3d(hipStream_t st, const float *arr, size_t nx, size_t ny, size_t L, size_t s,
                    float *out)
{
    const size_t pixels = (nx / s) * (ny / s);
    const bool vec = L % 4 == 0 && ((uintptr_t)arr | (uintptr_t)out) % 16 == 0;
    if (vec) THZ_LAUNCH(k_scale3d<true>, grid_1d(pixels * kWave, 256, kNumCU * 8), 256, 0, st, arr, nx, ny, L, s, out);
    else THZ_LAUNCH(k_scale3d<false>, grid_1d(pixels * kWave, 256, kNumCU * 8), 256, 0, st, arr, nx, ny, L, s, out);
}

void launch_tilt(hipStream_t st, size_t npix, int nt_in, int nt_out, const float *in,
                 const float *taper, const int *insert_index, float *out)
{
    THZ_LAUNCH(k_tilt, grid_1d(npix * kWave, 256, kNumCU * 8), 256, 0, st, npix, nt_in, nt_out, in, taper,
               insert_index, out);
}

// ---- deconvolution launchers
static inline void dc_geometry(const PlanDev &P, size_t npix, int bufs, unsigned *grid,
                               unsigned *block, size_t *lds)
{
    const size_t per_wave = (size_t)bufs * P.buf_entries * sizeof(c32);
    unsigned wpb = (unsigned)(kLdsBytesPerCU / per_wave);
    if (wpb > 4) wpb = 4;
    if (wpb < 1) wpb = 1;
    *block = wpb * kWave;
    *lds = per_wave * wpb;
    size_t per_cu = kLdsBytesPerCU / *lds;
    if (per_cu < 1) per_cu = 1;
    if (per_cu * wpb > 32) per_cu = 32 / wpb;
    size_t g = (npix + wpb - 1) / wpb;
    if (g > (size_t)kNumCU * per_cu) g = (size_t)kNumCU * per_cu;
    if (g < 1) g = 1;
    *grid = (unsigned)g;
}

void launch_dc_filter_spectra(hipStream_t st, const float *filters, int n_bands, int n_taps, const double *cs,
                              const double *sn, unsigned M, unsigned nk, c32 *H)
{
    THZ_LAUNCH(k_dc_filter_spectra, ((unsigned)n_bands * nk + 255) / 256, 256, 0, st, filters, n_bands, n_taps, cs, sn, M,
               nk, H);
}

// The forward transform of the zero-padded traces on the F core (round 3; k_dc_fft above is the generic LDS transform:
// 2.4 ms at 512 x 512 x 1001 for 3.2 GB of traffic): samples beyond nt enter as zeros — whole pass-1 blocks of them are
// not loaded at all — and the spectrum leaves through the F kernels' own epilogue (split in place, 16-byte stores).
// LDS: [T1][T2][ones: N + 4 floats][w2n head][wg][per wave: N + 2].
template <class PL>
__global__ __launch_bounds__(256) void k_dc_fft_f(FTables T, size_t npix, int nt, const float *__restrict__ in,
                                                  cx *__restrict__ spec)
{
    THZ_DYN_LDS(lds);
    constexpr int N = PL::N, R1 = PL::R1, C1 = PL::C1, M1 = PL::M1;
    constexpr int ONES = (N + 4) / 2;  // cx entries
    const int lane = lane_id();
    const int wib = (int)(threadIdx.x >> 6), wpb = (int)(blockDim.x >> 6);
    cx *t1 = reinterpret_cast<cx *>(lds);
    cx *t2 = t1 + PL::T1_ENTRIES;
    float *ones_s = reinterpret_cast<float *>(t2 + PL::T2_ENTRIES);
    cx *w2n_s = t2 + PL::T2_ENTRIES + ONES;
    cx *wg_s = w2n_s + PL::W2N_HEAD;
    cx *buf = wg_s + PL::WG_ENTRIES + (size_t)wib * PL::WAVE_ENTRIES;
    for (int i = (int)threadIdx.x; i < PL::W2N_HEAD; i += (int)blockDim.x) w2n_s[i] = f_stage_w2n(T.w2n[i]);
    if ((int)threadIdx.x < R1) wg_s[threadIdx.x] = T.w2n[M1 * (int)threadIdx.x];
    for (int i = (int)threadIdx.x; i < PL::T1_ENTRIES; i += (int)blockDim.x) t1[i] = T.t1[i];
    for (int i = (int)threadIdx.x; i < PL::T2_ENTRIES; i += (int)blockDim.x) t2[i] = T.t2[i];
    for (int i = (int)threadIdx.x; i < 2 * ONES; i += (int)blockDim.x) ones_s[i] = 1.0f;
    __syncthreads();
    FAddr<PL> ad;
    ad.init(lane);
    FArgs A{};
    A.npix = npix;
    A.fft_out = spec;
    for (size_t p = (size_t)blockIdx.x * wpb + wib; p < npix; p += (size_t)gridDim.x * wpb) {
        const float *x = in + p * (size_t)nt;
        ad.refresh();
        cx r[C1][R1];
        // z[n] = x[2 n] + i x[2 n + 1], n = M1 j1 + C1 lane + c: block j1 holds the samples [2 M1 j1, 2 M1 (j1 + 1))
#pragma unroll
        for (int j1 = 0; j1 < R1; ++j1) {
            const int s0 = 2 * (M1 * j1 + C1 * lane);
            if (2 * M1 * j1 >= nt) {  // wave-uniform: nothing of this block exists
#pragma unroll
                for (int c = 0; c < C1; ++c) r[c][j1] = cx{0.0f, 0.0f};
            } else if (2 * M1 * (j1 + 1) <= nt) {  // wave-uniform: all of it does
                if constexpr (C1 == 2) {
                    float a, b, c_, d;
                    load_f4(x + s0, a, b, c_, d);
                    r[0][j1] = cx{a, b};
                    r[1][j1] = cx{c_, d};
                } else {
                    r[0][j1] = cx{x[s0], x[s0 + 1]};
                }
            } else {
#pragma unroll
                for (int c = 0; c < C1; ++c) {
                    const int i0 = s0 + 2 * c;
                    r[c][j1] = cx{i0 < nt ? x[i0] : 0.0f, i0 + 1 < nt ? x[i0 + 1] : 0.0f};
                }
            }
        }
        f_core_pass1<PL>(r, buf, t1, ad, lane);
        f_core_pass23<PL>(buf, t2, ad, lane);
        f_spectrum_epilogue<PL, false>(buf, w2n_s, wg_s, ones_s, p, A, lane);
        wave_sync();
    }
}

template <class PL>
static void launch_dc_fft_f(hipStream_t st, const PlanDev &P, size_t npix, int nt, const float *in, c32 *spec)
{
    FTables T{reinterpret_cast<const cx *>(P.f_t1), reinterpret_cast<const cx *>(P.f_t2),
              reinterpret_cast<const cx *>(P.f_w2n)};
    const unsigned wpb = 4;
    const size_t lds = (size_t)(PL::T1_ENTRIES + PL::T2_ENTRIES + (PL::N + 4) / 2 + PL::W2N_HEAD + PL::WG_ENTRIES
                                + wpb * PL::WAVE_ENTRIES) * sizeof(cx);
    size_t per_cu = kLdsBytesPerCU / lds;
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 4) per_cu = 4;
    size_t g = (npix + wpb - 1) / wpb;
    if (g > (size_t)kNumCU * per_cu) g = (size_t)kNumCU * per_cu;
    allow_dynamic_lds(k_dc_fft_f<PL>, lds);
    THZ_LAUNCH((k_dc_fft_f<PL>), (unsigned)g, wpb * kWave, lds, st, T, npix, nt, in, reinterpret_cast<cx *>(spec));
}

void launch_dc_fft(hipStream_t st, const PlanDev &P, size_t npix, int nt, const float *in, c32 *spec)
{
    // THZ_DC_FFT_OLD=1 (developer knob, A/B runs): the generic LDS transform for every padded length
    static const bool old_form = getenv("THZ_DC_FFT_OLD") != nullptr;
    if (!old_form && P.f_t1 && P.f_t2 && P.f_w2n && nt <= P.nt) {  // the F core's tables came with the plan
        switch (P.nt) {
        case 4096: launch_dc_fft_f<FPlan4096>(st, P, npix, nt, in, spec); return;
        case 2048: launch_dc_fft_f<FPlan2048>(st, P, npix, nt, in, spec); return;
        case 1024: launch_dc_fft_f<FPlan1024>(st, P, npix, nt, in, spec); return;
        default: break;
        }
    }
    unsigned grid, block;
    size_t lds;
    wave_launch_geometry(P, npix, &grid, &block, &lds);
    allow_dynamic_lds(k_dc_fft, lds);
    THZ_LAUNCH(k_dc_fft, grid, block, lds, st, P, npix, nt, in, spec);
}

// The recombination on the F core: per pixel the gain-weighted sum of the filter spectra is built in
// registers (bin by bin in the spectrum's register layout), multiplied into the spectrum, and ONE
// inverse transform gives the "same" slice of the output trace and its intensity.
// PX pixels per wave share every band's row of H (round 3: at one pixel per wave the rows were 52 GB of L2 reads per
// call at 512 x 512 pixels — 2.7 ms, which is what L2 delivers); the sums are packed FMAs over {re, im} pairs.
template <class PL, int PX>
__global__ __launch_bounds__(512) void k_dc_combine_f(FTables T, size_t npix, int nt, int n_bands, int shift,
                                                      const cx *__restrict__ spec, const cx *__restrict__ H,
                                                      const float *__restrict__ gain, float *__restrict__ out,
                                                      float *__restrict__ img)
{
    THZ_DYN_LDS(lds);
    constexpr int N = PL::N, R1 = PL::R1, C1 = PL::C1;
    const int nf = N + 1;
    const int lane = lane_id();
    const int wib = (int)(threadIdx.x >> 6), wpb = (int)(blockDim.x >> 6);
    cx *t1 = reinterpret_cast<cx *>(lds);
    cx *t2 = t1 + PL::T1_ENTRIES;
    cx *w2n_s = t2 + PL::T2_ENTRIES;
    cx *wg_s = w2n_s + PL::W2N_HEAD;
    cx *buf = wg_s + PL::WG_ENTRIES + (size_t)wib * PL::WAVE_ENTRIES;
    for (int i = (int)threadIdx.x; i < PL::W2N_HEAD; i += (int)blockDim.x) w2n_s[i] = f_stage_w2n(T.w2n[i]);
    if ((int)threadIdx.x < R1) wg_s[threadIdx.x] = T.w2n[PL::M1 * (int)threadIdx.x];
    for (int i = (int)threadIdx.x; i < PL::T1_ENTRIES; i += (int)blockDim.x) t1[i] = T.t1[i];
    for (int i = (int)threadIdx.x; i < PL::T2_ENTRIES; i += (int)blockDim.x) t2[i] = T.t2[i];
    __syncthreads();
    FAddr<PL> ad;
    ad.init(lane);
    const int s2 = (lane >> 4) & 3;
    const int sb2 = (2 * lane) ^ (s2 & 2);
    const bool swap2 = (s2 & 1) != 0;
    const int sb1a = nat(lane), sb1b = nat(kWave + lane) - kWave;
    for (size_t p0 = ((size_t)blockIdx.x * wpb + wib) * PX; p0 < npix; p0 += (size_t)gridDim.x * wpb * PX) {
        cx hc[PX][R1][C1];
        float hc_nyq[PX], hs[R1][2 * C1];
#pragma unroll
        for (int q = 0; q < PX; ++q) {
            hc_nyq[q] = 0.0f;
#pragma unroll
            for (int j = 0; j < R1; ++j)
#pragma unroll
                for (int c = 0; c < C1; ++c) hc[q][j][c] = cx{0.0f, 0.0f};
        }
        f_load_spec<PL>(H, lane, hs);
        float h_nyq = H[N].x;
#pragma unroll 1
        for (int b = 0; b < n_bands; ++b) {
            float g[PX];
#pragma unroll
            for (int q = 0; q < PX; ++q) g[q] = gain[(size_t)b * npix + (p0 + q < npix ? p0 + q : p0)];
#pragma unroll
            for (int q = 0; q < PX; ++q) {
#pragma unroll
                for (int j = 0; j < R1; ++j)
#pragma unroll
                    for (int c = 0; c < C1; ++c) hc[q][j][c] += cx{g[q], g[q]} * cx{hs[j][2 * c], hs[j][2 * c + 1]};
                hc_nyq[q] += g[q] * h_nyq;
            }
            if (b + 1 < n_bands) {
                f_load_spec<PL>(H + (size_t)(b + 1) * nf, lane, hs);
                h_nyq = H[(size_t)(b + 1) * nf + N].x;
            }
        }
#pragma unroll
        for (int q = 0; q < PX; ++q) {
        const size_t p = p0 + q;
        if (p >= npix) break;  // wave-uniform
        f_load_spec<PL>(spec + p * nf, lane, hs);  // the pixel's spectrum
        const float x_nyq = spec[p * nf + N].x;
        ad.refresh();
#pragma unroll
        for (int j = 0; j < R1; ++j) {
            if constexpr (C1 == 2) {
                const cx e0 = cx_mul(cx{hs[j][0], hs[j][1]}, hc[q][j][0]);
                const cx e1 = cx_mul(cx{hs[j][2], hs[j][3]}, hc[q][j][C1 - 1]);
                st2(buf + sb2 + 2 * kWave * j, swap2 ? e1 : e0, swap2 ? e0 : e1);
            } else {
                buf[((j & 1) ? sb1b : sb1a) + kWave * j] = cx_mul(cx{hs[j][0], hs[j][1]}, hc[q][j][0]);
            }
        }
        if (lane == 0) buf[N] = cx{x_nyq * hc_nyq[q], 0.0f};
        wave_sync();
        cx r[C1][R1];
        f_inverse_input<PL, false>(buf, w2n_s, wg_s, nullptr, lane, r);
        wave_sync();
        f_core_pass1<PL>(r, buf, t1, ad, lane);
        f_core_pass23<PL>(buf, t2, ad, lane);
        float acc = 0.0f;
        float *o = out + p * (size_t)nt;
#pragma unroll
        for (int j = 0; j < R1; ++j) {
            float v[2 * C1];
            if constexpr (C1 == 2) {
                const cx2 rr = ld2(buf + sb2 + 2 * kWave * j);
                const cx e0 = swap2 ? rr.b : rr.a, e1 = swap2 ? rr.a : rr.b;
                v[0] = e0.y; v[1] = e0.x; v[2] = e1.y; v[3] = e1.x;
            } else {
                const cx rr = buf[((j & 1) ? sb1b : sb1a) + kWave * j];
                v[0] = rr.y; v[1] = rr.x;
            }
            const int t0 = 2 * C1 * (kWave * j + lane) - shift;
#pragma unroll
            for (int i = 0; i < 2 * C1; ++i)
                if (t0 + i >= 0 && t0 + i < nt) {
                    o[t0 + i] = v[i];
                    acc += v[i] * v[i];
                }
        }
        if (img) {
            acc = wave_reduce_add(acc);
            if (lane == 0) img[p] = acc;
        }
        wave_sync();
        }
    }
}

template <class PL>
static void launch_dc_energy_f(hipStream_t st, const PlanDev &P, size_t npix, int nt, int n_bands, int shift,
                               const c32 *spec, const c32 *H, float *energy)
{
    FTables T{reinterpret_cast<const cx *>(P.f_t1), reinterpret_cast<const cx *>(P.f_t2),
              reinterpret_cast<const cx *>(P.f_w2n)};
    // N <= 1024 (nt + 498 <= 2048: every scan of the sample data): four waves per block, three blocks per CU
    // (k_dc_energy_f3).  THZ_DC_ENERGY_OLD=1 (developer knob) keeps round 2's eight-wave blocks for A/B runs.
    static const bool old_form = getenv("THZ_DC_ENERGY_OLD") != nullptr;
    if constexpr (PL::N <= 1024) if (!old_form) {
        const unsigned w3 = 4;
        const size_t lds3 = (size_t)(PL::T1_ENTRIES + PL::T2_ENTRIES + PL::W2N_HEAD + PL::WG_ENTRIES + w3 * PL::WAVE_ENTRIES) * sizeof(cx);
        size_t per_cu3 = kLdsBytesPerCU / lds3;
        const size_t occ3 = PL::N <= 512 ? 4 : 3;  // waves per SIMD the kernel is compiled for = blocks of four waves per CU
        if (per_cu3 > occ3) per_cu3 = occ3;
        if (per_cu3 < 1) per_cu3 = 1;
        size_t g3 = (npix + w3 - 1) / w3;
        if (g3 > (size_t)kNumCU * per_cu3) g3 = (size_t)kNumCU * per_cu3;
        allow_dynamic_lds(k_dc_energy_f3<PL>, lds3);
        THZ_LAUNCH((k_dc_energy_f3<PL>), (unsigned)g3, w3 * kWave, lds3, st, T, npix, nt, n_bands, shift,
                   reinterpret_cast<const cx *>(spec), reinterpret_cast<const cx *>(H), energy);
        return;
    }
    const unsigned wpb = 8;
    const size_t lds = (size_t)(PL::T1_ENTRIES + PL::T2_ENTRIES + PL::W2N_HEAD + PL::WG_ENTRIES + wpb * PL::WAVE_ENTRIES)
                       * sizeof(cx);
    size_t per_cu = kLdsBytesPerCU / lds;
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 2) per_cu = 2;
    size_t g = (npix + wpb - 1) / wpb;
    if (g > (size_t)kNumCU * per_cu) g = (size_t)kNumCU * per_cu;
    allow_dynamic_lds(k_dc_energy_f<PL>, lds);
    THZ_LAUNCH((k_dc_energy_f<PL>), (unsigned)g, wpb * kWave, lds, st, T, npix, nt, n_bands, shift,
               reinterpret_cast<const cx *>(spec), reinterpret_cast<const cx *>(H), energy);
}

void launch_dc_energy(hipStream_t st, const PlanDev &P, size_t npix, int nt, int n_bands, int shift,
                      const c32 *spec, const c32 *H, float *energy)
{
    if (P.f_t1 && P.f_t2 && P.f_w2n && nt + shift <= P.nt) {  // the F core's tables came with the plan
        switch (P.nt) {
        case 4096: launch_dc_energy_f<FPlan4096>(st, P, npix, nt, n_bands, shift, spec, H, energy); return;
        case 2048: launch_dc_energy_f<FPlan2048>(st, P, npix, nt, n_bands, shift, spec, H, energy); return;
        case 1024: launch_dc_energy_f<FPlan1024>(st, P, npix, nt, n_bands, shift, spec, H, energy); return;
        default: break;
        }
    }
    unsigned grid, block;
    size_t lds;
    dc_geometry(P, npix, 3, &grid, &block, &lds);
    allow_dynamic_lds(k_dc_energy, lds);
    THZ_LAUNCH(k_dc_energy, grid, block, lds, st, P, npix, nt, n_bands, shift, spec, H, energy);
}

// ---- the band energies in Parseval form (k_dc_energy_pv)
bool dc_energy_pv_supported(size_t M, int n_taps)
{
    const int s = (n_taps - 1) / 2;
    return (n_taps & 1) && s >= 1 && 2 * s - 1 <= 512 && (M == 1024 || M == 2048 || M == 4096 || M == 8192 || M == 16384);
}

void launch_dc_pv_tables(hipStream_t st, int n_bands, int nk, int gstride, size_t M, const c32 *H, const c32 *hht, float *g,
                         c32 *hpm)
{
    const int total = n_bands * (gstride + 512);
    THZ_LAUNCH(k_dc_pv_tables, (unsigned)((total + 255) / 256), 256, 0, st, n_bands, nk, gstride, (float)M, H, hht, g,
               reinterpret_cast<cx2 *>(hpm));
}

template <int NG, int PX>
static void launch_dc_energy_pv_n(hipStream_t st, const DcPvDev &T, size_t npix, int nt, int n_bands, int shift, int nk,
                                  const float *in, const c32 *spec, float *energy)
{
    {
        const size_t waves = (npix + PX - 1) / PX;
        size_t g = (waves + 3) / 4;
        if (g > (size_t)kNumCU * 8) g = (size_t)kNumCU * 8;
        if (g_grid_cap_override > 0 && g > (size_t)g_grid_cap_override) g = (size_t)g_grid_cap_override;
        THZ_LAUNCH((k_dc_energy_full<NG, PX>), (unsigned)g, 256, 0, st, T, npix, n_bands, nk, reinterpret_cast<const cx *>(spec),
                   energy);
    }
    // THZ_DC_EDGE_WAVES=4 (developer knob): blocks of four waves, four per CU, instead of eight-wave blocks, two per CU
    static const bool four = [] { const char *e = getenv("THZ_DC_EDGE_WAVES"); return e && e[0] == '4'; }();
    auto go = [&](auto kernel, int waves) {
        const size_t lds = waves == 8 ? DcPvLds<8>::bytes() : DcPvLds<4>::bytes();
        const size_t per_cu = 16 / waves;  // sixteen waves per CU: the kernel is compiled for four per SIMD
        size_t g = (npix + waves - 1) / waves;
        if (g > (size_t)kNumCU * per_cu) g = (size_t)kNumCU * per_cu;
        if (g_grid_cap_override > 0 && g > (size_t)g_grid_cap_override) g = (size_t)g_grid_cap_override;  // tests: several batches per block
        allow_dynamic_lds(kernel, lds);
        THZ_LAUNCH(kernel, (unsigned)g, (unsigned)(waves * kWave), lds, st, T, npix, nt, n_bands, shift, in, energy);
    };
    if (shift == 249) {
        if (four) go(k_dc_energy_edges<249, 4>, 4);
        else go(k_dc_energy_edges<249, 8>, 8);
    } else {
        go(k_dc_energy_edges<0, 8>, 8);
    }
}

int dc_pv_gstride(int nk) { return nk + 3; }  // [bins 0 .. nk - 1)[Nyquist][0 0 0], see DcPvLds

void launch_dc_energy_pv(hipStream_t st, const DcPvTables &Tb, size_t npix, int nt, int n_bands, int shift, int nk,
                         const float *in, const c32 *spec, float *energy)
{
    const DcPvDev T{reinterpret_cast<const cx *>(Tb.t1), reinterpret_cast<const cx *>(Tb.t2),
                    reinterpret_cast<const cx2 *>(Tb.hpm), Tb.g, Tb.gstride};
    if (Tb.gstride == dc_pv_gstride(nk)) switch (nk) {
        case 513: launch_dc_energy_pv_n<2, 4>(st, T, npix, nt, n_bands, shift, nk, in, spec, energy); return;
        case 1025: launch_dc_energy_pv_n<4, 4>(st, T, npix, nt, n_bands, shift, nk, in, spec, energy); return;
        case 2049: launch_dc_energy_pv_n<8, 2>(st, T, npix, nt, n_bands, shift, nk, in, spec, energy); return;
        // padded lengths without an F core (nt > 3598): the edges do not depend on M, the Parseval sums only read more bins
        case 4097: launch_dc_energy_pv_n<16, 1>(st, T, npix, nt, n_bands, shift, nk, in, spec, energy); return;
        case 8193: launch_dc_energy_pv_n<32, 1>(st, T, npix, nt, n_bands, shift, nk, in, spec, energy); return;
        default: break;
        }
    fprintf(stderr, "thzgpu: launch_dc_energy_pv: no kernel for %d bins, row stride %d (dc_energy_pv_supported, dc_pv_gstride)\n",
            nk, Tb.gstride);
    abort();
}

template <class PL>
static void launch_dc_combine_f(hipStream_t st, const PlanDev &P, size_t npix, int nt, int n_bands, int shift,
                                const c32 *spec, const c32 *H, const float *gain, float *out, float *img)
{
    const unsigned wpb = 8;
    const size_t lds = (size_t)(PL::T1_ENTRIES + PL::T2_ENTRIES + PL::W2N_HEAD + PL::WG_ENTRIES + wpb * PL::WAVE_ENTRIES)
                       * sizeof(cx);
    size_t per_cu = kLdsBytesPerCU / lds;
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 2) per_cu = 2;
    // pixels per wave: THZ_DC_COMBINE_PX (developer knob: 1, 2, 4); default by the plan's register budget
    static const int px_env = [] { const char *e = getenv("THZ_DC_COMBINE_PX"); return e ? atoi(e) : 0; }();
    constexpr int kDefaultPx = PL::N <= 1024 ? 2 : 1;
    const int px = (px_env == 1 || px_env == 2 || px_env == 4) ? px_env : kDefaultPx;
    size_t g = (npix + (size_t)wpb * px - 1) / ((size_t)wpb * px);
    if (g > (size_t)kNumCU * per_cu) g = (size_t)kNumCU * per_cu;
    FTables T{reinterpret_cast<const cx *>(P.f_t1), reinterpret_cast<const cx *>(P.f_t2),
              reinterpret_cast<const cx *>(P.f_w2n)};
    auto go = [&](auto kernel) {
        allow_dynamic_lds(kernel, lds);
        THZ_LAUNCH(kernel, (unsigned)g, wpb * kWave, lds, st, T, npix, nt, n_bands, shift,
                   reinterpret_cast<const cx *>(spec), reinterpret_cast<const cx *>(H), gain, out, img);
    };
    if constexpr (PL::N <= 1024) {
        if (px == 4) return go(k_dc_combine_f<PL, 4>);
        if (px == 2) return go(k_dc_combine_f<PL, 2>);
    }
    go(k_dc_combine_f<PL, 1>);
}

void launch_dc_combine(hipStream_t st, const PlanDev &P, size_t npix, int nt, int n_bands, int shift,
                       const c32 *spec, const c32 *H, const float *gain, float *out, float *img)
{
    if (P.f_t1 && P.f_t2 && P.f_w2n && nt + shift <= P.nt) {
        switch (P.nt) {
        case 4096: launch_dc_combine_f<FPlan4096>(st, P, npix, nt, n_bands, shift, spec, H, gain, out, img); return;
        case 2048: launch_dc_combine_f<FPlan2048>(st, P, npix, nt, n_bands, shift, spec, H, gain, out, img); return;
        case 1024: launch_dc_combine_f<FPlan1024>(st, P, npix, nt, n_bands, shift, spec, H, gain, out, img); return;
        default: break;
        }
    }
    unsigned grid, block;
    size_t lds;
    dc_geometry(P, npix, 2, &grid, &block, &lds);
    allow_dynamic_lds(k_dc_combine, lds);
    THZ_LAUNCH(k_dc_combine, grid, block, lds, st, P, npix, nt, n_bands, shift, spec, H, gain, out, img);
}

bool dc_combine_has_f_core(const PlanDev &P, int nt, int shift)
{
    return P.f_t1 && P.f_t2 && P.f_w2n && nt + shift <= P.nt && (P.nt == 4096 || P.nt == 2048 || P.nt == 1024);
}

bool dc_weight_spectra_supported(int n_bands) { return n_bands >= 1 && (size_t)n_bands * kDcWeightBins * sizeof(cx) <= 128 * 1024; }

void launch_dc_weight_spectra(hipStream_t st, size_t npix, size_t gain_stride, int nk, int n_bands, const c32 *spec,
                              const c32 *H, const float *gain, c32 *y)
{
    const size_t lds = (size_t)n_bands * kDcWeightBins * sizeof(cx);
    const unsigned gx = (unsigned)((nk + kDcWeightBins - 1) / kDcWeightBins);
    size_t per_cu = kLdsBytesPerCU / lds;
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    size_t gy = ((size_t)kNumCU * per_cu + gx - 1) / gx;
    const size_t need = (npix + 3) / 4;
    if (gy > need) gy = need;
    if (gy < 1) gy = 1;
    allow_dynamic_lds(k_dc_weight_spectra, lds);
    THZ_LAUNCH(k_dc_weight_spectra, gx * (unsigned)gy, 256, lds, st, npix, gain_stride, nk, n_bands, gx,
               reinterpret_cast<const cx *>(spec), reinterpret_cast<const cx *>(H), gain, reinterpret_cast<cx *>(y));
}

void launch_rl_init(hipStream_t st, const RlBand *d_bands, int n_bands, unsigned total_blocks,
                    size_t npix, const float *energy, float *ws)
{
    THZ_LAUNCH(k_rl_init, total_blocks, 256, 0, st, d_bands, n_bands, npix, energy, ws);
}

void launch_rl_step(hipStream_t st, const RlBand *d_bands, int n_bands, unsigned total_blocks,
                    const int *it_base, int iteration, int step, float *ws)
{
    THZ_LAUNCH(k_rl_step, total_blocks, 256, 0, st, d_bands, n_bands, it_base, iteration, step, ws);
}

size_t rl_tile_lds_bytes(int pr, int pc, bool separable)
{
    const bool turned = rl_turned(pr, pc);
    if (separable) return rl_sep_floats(pr, pc) * sizeof(float);
    if (!turned) return (kRlTilesPerBlock * rl_tile_floats(pr, pc, false) + rl_tap_floats(pr, pc)) * sizeof(float);
    return (rl_tile_floats(pr, pc, true) + rl_tap_floats(pr, pc) + (size_t)kRlSplit * 256) * sizeof(float);
}

unsigned rl_tile_block_count(int pr, int pc, unsigned n_tiles) { return rl_tile_blocks(rl_turned(pr, pc), n_tiles); }

// threads of a separable tile's block.  Measured (profiles/r03_rl_block_size.txt): while a launch needs several rounds
// of blocks on the chip (512 x 512 pixels: 2 128 tiles for 512 places of 1 024 threads) eight waves per tile put three
// tiles on a CU instead of two and the iterations take 24.0 ms instead of 26.2; once every tile has a place (128 x 128:
// 112 tiles) the sixteen-wave block is the shorter one (6.65 against 7.07 ms); four waves lose both ways.
// THZ_RL_SEP_THREADS (developer knob) overrides.
static int rl_sep_threads(unsigned total_tiles)
{
    static const int forced = [] {
        if (const char *e = getenv("THZ_RL_SEP_THREADS")) {
            const int v = atoi(e);
            if (v == 256 || v == 512 || v == 1024) return v;
        }
        return 0;
    }();
    if (forced) return forced;
    return total_tiles > 2u * kNumCU ? 512 : 1024;
}

void prepare_rl_step_tiled(int kind, size_t lds_bytes)
{
    if (kind == kRlSeparable) {
        allow_dynamic_lds(k_rl_step_sep<1024>, lds_bytes);
        allow_dynamic_lds(k_rl_step_sep<512>, lds_bytes);
        allow_dynamic_lds(k_rl_step_sep<256>, lds_bytes);
    }
    else if (kind == kRlWide) allow_dynamic_lds(k_rl_step_tiled<true>, lds_bytes);
    else allow_dynamic_lds(k_rl_step_tiled<false>, lds_bytes);
}

void launch_rl_step_tiled(hipStream_t st, int kind, const RlTileRef *d_tiles, unsigned total_tiles, size_t lds_bytes,
                          const int *it_base, int iteration, int step, float *ws)
{
    if (total_tiles == 0) return;
    if (kind == kRlSeparable) {
        const int nt_sep = rl_sep_threads(total_tiles);
        if (nt_sep == 256) THZ_LAUNCH(k_rl_step_sep<256>, total_tiles, 256, lds_bytes, st, d_tiles, it_base, iteration, step, ws);
        else if (nt_sep == 512) THZ_LAUNCH(k_rl_step_sep<512>, total_tiles, 512, lds_bytes, st, d_tiles, it_base, iteration, step, ws);
        else THZ_LAUNCH(k_rl_step_sep<1024>, total_tiles, 1024, lds_bytes, st, d_tiles, it_base, iteration, step, ws);
    }
    else if (kind == kRlWide) THZ_LAUNCH(k_rl_step_tiled<true>, total_tiles, kRlThreads, lds_bytes, st, d_tiles, it_base, iteration, step, ws);
    else THZ_LAUNCH(k_rl_step_tiled<false>, total_tiles, kRlNarrowThreads, lds_bytes, st, d_tiles, it_base, iteration, step, ws);
}

void launch_dc_gain(hipStream_t st, const RlBand *d_bands, int n_bands, size_t npix,
                    const float *energy, const float *ws, float *gain)
{
    THZ_LAUNCH(k_dc_gain, grid_1d((size_t)n_bands * npix, 256, kNumCU * 8), 256, 0, st, d_bands,
               n_bands, npix, energy, ws, gain);
}

void launch_synth(hipStream_t st, float *out, size_t ntraces, int nt, uint64_t first_trace,
                  const float *time, uint32_t seed, int subtract_bias)
{
    THZ_LAUNCH(k_synth, grid_1d(ntraces * nt, 256, kNumCU * 16), 256, 0, st, out, ntraces, nt,
               first_trace, time, seed, subtract_bias);
}

}  // namespace thz
