// fft_fb.hpp — "FB" kernels: the fused default chain for trace lengths that are not a power of
// two (real instrument scans: nt ~ 1000), as a chirp-z (Bluestein) transform over the
// register-resident complex FFT core of the F family (fft_f.hpp).
//
//   X[k] = w[k] * sum_n (x[n] w[n]) conj(w)[k-n],   w[n] = exp(-i pi n^2 / nt)
//
// i.e. one circular convolution of length M >= 2 nt - 1 (M = 512 / 1024 / 2048 = the complex
// sizes of the three F plans) = two M-point FFTs through f_core_pass1/f_core_pass23; the inverse
// real transform is the same machinery applied to conj(Y) (y = Re DFT(conj Y) / nt).  Four
// M-point FFTs per trace: the kernel is VALU-bound, not HBM-bound (algorithmic bytes are
// 16 nt + 20 per trace as for the power-of-two chain).
//
// Same parity rules as the G kernels it replaces for these lengths (kernels.hip k_fft_fwd /
// k_fft_inv): DC (and Nyquist for even nt) imaginary parts are zero, phases are taken before the
// band-pass, the masked spectrum feeds the inverse.
#pragma once

namespace thz {

struct FBArgs {
    size_t npix;
    int nt, nf;
    const float *in;        // (npix, nt)
    const float *pre_win;   // nt or nullptr
    const float *mask;      // nf (never null: the plan's ones vector stands in)
    const float *post_win;  // nt or nullptr
    cx *fft_out;            // (npix, nf)
    float *amp_out, *ph_out;
    float *data_out;        // (npix, nt)
    float *img;             // npix or nullptr
    const cx *w;            // nt: exp(-i pi n^2 / nt)
    const cx *bf;           // M : FFT_M(b) / M
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

template <class P>
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
    const float kPi = 3.14159274101257324219f, kTwoPi = 2.0f * kPi;
    const float fnt = (float)L;
    const int n_groups = (nf + 255) / 256;  // epilogue groups of 256 bins: bin = 256 g + 4 lane + c
    int fb4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) fb4[c] = nat(4 * lane + c);  // nat(256 g + 4 lane + c) = 256 g + fb4[c]
    const size_t stride = (size_t)gridDim.x * wpb;

    for (size_t p = (size_t)blockIdx.x * wpb + wib; p < A.npix; p += stride) {
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

        // ---- a[n] = x[n] pre[n] w[n] (zero from nt on), in the core's input layout.  Branch-free:
        // indices are clamped and the value selected, so that the loads of a trace are issued
        // together instead of one exec-masked round trip each
        {
            const float *x = A.in + p * (size_t)L;
            constexpr int H = R1 / 2;  // two batches: 3 registers per element in flight, not for all 32
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float xv[C1][H];
                cx wv[C1][H];
#pragma unroll
                for (int j = 0; j < H; ++j) {
#pragma unroll
                    for (int c = 0; c < C1; ++c) {
                        const int n = M1 * (H * h + j) + lb + c;
                        const unsigned nn = (unsigned)(n < L ? n : L - 1);  // unsigned: SGPR base + 32-bit offset
                        xv[c][j] = ld_off(x, nn);
                        wv[c][j] = ld_off(wl, nn);
                    }
                }
#pragma unroll
                for (int j = 0; j < H; ++j) {
#pragma unroll
                    for (int c = 0; c < C1; ++c) {
                        const int n = M1 * (H * h + j) + lb + c;
                        const int nn = n < L ? n : L - 1;
                        const float t = n < L ? xv[c][j] * pre_l[nn] : 0.0f;
                        r[c][H * h + j] = cx{t * wv[c][j].x, t * wv[c][j].y};
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

        // ---- spectrum epilogue: X[k] = w[k] * c[k], k < nf; bins 256 g + 4 lane + c
        {
            float carry = 0.0f, prev_tail = 0.0f, first = 0.0f;
            for (int g = 0; g < n_groups; ++g) {
                const int k0 = 256 * g + lb4;
                cx X[4];
                float m[4];
                bool ok[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int k = k0 + c;
                    ok[c] = k < nf;
                    const int kc = ok[c] ? k : nf - 1;
                    const cx s = buf[nat(kc)];
                    X[c] = cx_mul(cx{s.y, s.x}, ld_off(wl, (unsigned)kc));
                    m[c] = mask_l[kc];
                    if (kc == 0 || ((L & 1) == 0 && kc == nf - 1)) X[c].y = 0.0f;  // real input
                }
                float a[4], ph[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    a[c] = fast_sqrt(fmaf(X[c].x, X[c].x, X[c].y * X[c].y)) * m[c];
                    ph[c] = fast_atan2f(X[c].y, X[c].x);
                }
                // numpy_unwrap over ascending bins (same two-level scan as the F epilogue)
                if (g == 0) first = wave_bcast<0>(ph[0]);
                float prev = wave_shr1(ph[3]);
                if (lane == 0) prev = prev_tail;
                float s_[4], run = 0.0f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float d = ph[c] - (c == 0 ? prev : ph[c - 1]);
                    d += (d > kPi) ? -kTwoPi : ((d < -kPi) ? kTwoPi : 0.0f);
                    if ((g == 0 && c == 0 && lane == 0) || !ok[c]) d = 0.0f;
                    run += d;
                    s_[c] = run;
                }
                const float incl = wave_scan_add(run);
                const float excl = wave_shr1(incl);
                const float base = carry + excl;
                carry += wave_bcast<kWave - 1>(incl);
                prev_tail = wave_bcast<kWave - 1>(ph[3]);
                cx *f = A.fft_out + p * (size_t)nf + k0;
                float *ao = A.amp_out + p * (size_t)nf + k0, *po = A.ph_out + p * (size_t)nf + k0;
                if (ok[3]) {
                    store_f4(reinterpret_cast<float *>(f), X[0].x * m[0], X[0].y * m[0], X[1].x * m[1], X[1].y * m[1]);
                    store_f4(reinterpret_cast<float *>(f) + 4, X[2].x * m[2], X[2].y * m[2], X[3].x * m[3],
                             X[3].y * m[3]);
                    store_f4(ao, a[0], a[1], a[2], a[3]);
                    store_f4(po, first + (base + s_[0]), first + (base + s_[1]), first + (base + s_[2]),
                             first + (base + s_[3]));
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (ok[c]) {
                            f[c] = cx{X[c].x * m[c], X[c].y * m[c]};
                            ao[c] = a[c];
                            po[c] = first + (base + s_[c]);
                        }
                }
                // the masked spectrum stays in the wave's slice (own slots) for the inverse
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (ok[c]) buf[256 * g + fb4[c]] = cx{X[c].x * m[c], X[c].y * m[c]};
            }
        }
        wave_sync();

        // ---- inverse: a'[n] = conj(Yfull[n]) w[n];  Yfull[n] = Y[n] (n <= nt/2), conj(Y[nt-n]) above
        {
            const int half = L / 2;
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
                        const int kk = nn <= half ? nn : L - nn;
                        cx y = buf[nat(kk)];
                        y.y = nn <= half ? -y.y : y.y;  // conj(Y[n]); above: conj(conj(Y[nt-n])) = Y[nt-n]
                        const cx v = cx_mul(y, wv[c][j]);
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

        // ---- y[n] = Re(w[n] * c'[n]) / nt * post[n]; image = sum y^2
        {
            float *o = A.data_out + p * (size_t)L;
            float acc = 0.0f;
#pragma unroll 4
            for (int n = lb1; n < L; n += kWave) {
                const cx s = buf[nat(n)];
                const cx wv = ld_off(wl, (unsigned)n);
                // Re( swap(s) * w ) = s.y w.x - s.x w.y
                float v = (s.y * wv.x - s.x * wv.y) / fnt;
                v *= post_l[n];
                o[n] = v;
                acc += v * v;
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
