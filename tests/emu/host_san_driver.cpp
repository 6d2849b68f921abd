// host_san_driver.cpp — the O(Nt) host math of the engine (csrc/host_windows.cpp, csrc/deconv_host.cpp:
// windows, band-pass index rules, tilt plan, reference alignment, optical properties, FIR bank, PSF
// evaluation) under AddressSanitizer + UBSan, on ordinary and on edge inputs (shortest axes, bounds outside
// the axis, zero widths, PSFs wider than the image).  TEST INFRASTRUCTURE ONLY; see tests/test_sanitizers.py.
#include "deconv_host.hpp"
#include "host_windows.hpp"

#include <cmath>
#include <cstdio>
#include <vector>

using namespace thz;

static std::vector<float> axis(size_t n, float t0, float dt)
{
    std::vector<float> t(n);
    for (size_t i = 0; i < n; ++i) t[i] = t0 + dt * (float)i;
    return t;
}

int main()
{
    for (size_t nt : {2, 3, 7, 64, 1001, 4096}) {
        const auto t = axis(nt, 1000.0f, 0.05f);
        std::vector<float> w(nt);
        for (int type = 0; type < 5; ++type) fft_window(type, t.data(), nt, 1.0f, 7.0f, w.data());
        adapted_blackman(t.data(), nt, 0.0f, 0.0f, w.data());
        adapted_blackman(t.data(), nt, 1e6f, 1e6f, w.data());
        for (double lo : {-1e9, 999.0, 1000.0, 1001.0, 1e9})
            for (double hi : {-1e9, 1000.0, 1002.0, 1e9}) {
                double l = lo, h = hi;
                int64_t a = 0, b = 0;
                td_bandpass(t.data(), nt, &l, &h, 2.0, w.data(), &a, &b);
                td_bandpass(t.data(), nt, &l, &h, 0.0, w.data(), nullptr, nullptr);
            }
        const size_t nf = nt / 2 + 1;
        std::vector<float> f(nf), m(nf);
        for (size_t i = 0; i < nf; ++i) f[i] = (float)i / (t[nt - 1] - t[0]);
        for (double lo : {-5.0, 0.0, 0.2, 50.0})
            for (double hi : {-1.0, 0.1, 5.0, 1e6}) {
                int64_t a = 0, b = 0;
                fd_bandpass(f.data(), nf, lo, hi, 0.1, m.data(), &a, &b);
                fd_bandpass(f.data(), nf, lo, hi, 0.0, m.data(), nullptr, nullptr);
            }
        const float lines[3] = {0.557f, 0.752f, 1.097f};
        water_line_mask(f.data(), nf, lines, 3, 0.01f, m.data());
        water_line_mask(f.data(), nf, nullptr, 0, 0.01f, m.data());
        std::vector<float> ref(2 * nf, 0.5f), wf(2 * nf);
        wiener_filter(ref.data(), nf, 1e-3f, wf.data());
        // tilt: none, small, large
        for (double deg : {0.0, 0.5, 3.0}) {
            const size_t nx = 5, ny = 4;
            const size_t steps = tilt_plan(t.data(), nt, nx, ny, deg, -deg, 0.5f, 0.25f, nullptr, nullptr);
            std::vector<float> nt2(nt + 2 * steps);
            std::vector<int32_t> ins(nx * ny);
            tilt_plan(t.data(), nt, nx, ny, deg, -deg, 0.5f, 0.25f, nt2.data(), ins.data());
        }
        // reference pulse on a shifted / shorter / longer axis
        for (size_t nref : {(size_t)2, nt / 2 + 2, nt, nt + 37}) {
            for (float off : {-30.0f, 0.0f, 0.013f, 40.0f}) {
                const auto rt = axis(nref, 1000.0f + off, 0.05f);
                std::vector<float> rs(nref, 1.0f), out(nt), win(nt);
                align_reference(t.data(), nt, rt.data(), rs.data(), nref, out.data());
                reference_window(0, rt.data(), nref, 1.0f, 7.0f, nt, win.data());
            }
        }
        std::vector<float> amp(nf, 1.0f), ph(nf, 0.1f), n(nf), al(nf), ka(nf);
        optical_properties(amp.data(), ph.data(), amp.data(), ph.data(), f.data(), nf, 1.0f, n.data(), al.data(), ka.data());
        optical_properties(amp.data(), ph.data(), m.data(), ph.data(), f.data(), nf, 0.0f, n.data(), al.data(), ka.data());
    }
    for (int radius : {0, 1, 12, 40}) {
        std::vector<float> k((size_t)2 * radius + 1);
        gaussian_kernel1d(6.0f, radius, k.data());
        gaussian_kernel1d(0.0f, radius, k.data());
    }
    {
        std::vector<unsigned long long> hist(2048, 0);
        hist[7] = 3; hist[2047] = 1;
        int bin = 0;
        unsigned long long rem = 0;
        for (unsigned long long k : {1ull, 2ull, 4ull, 5ull, 1000ull}) select_step(hist.data(), 2048, k, &bin, &rem);
    }
    // FIR bank and PSF evaluation: a small hand-made PSF (2 ... 6 knots), bands from far below to far above the knots
    for (size_t nk : {(size_t)0, (size_t)2, (size_t)6}) {
        std::vector<float> kn(nk), va(nk), a(nk ? nk - 1 : 0), b(a.size()), c(a.size()), d(a.size());
        for (size_t i = 0; i < nk; ++i) { kn[i] = 0.2f + 0.3f * (float)i; va[i] = 0.01f * (float)i; }
        for (size_t i = 0; i + 1 < nk; ++i) { a[i] = va[i]; b[i] = 0.03f; c[i] = 0.001f; d[i] = -0.0005f; }
        thz_spline s{kn.data(), va.data(), a.data(), b.data(), c.data(), d.data(), nk};
        thz_psf P{};
        P.wx_fit = thz_hybrid_fit{0.86f, 0.17f, s};
        P.wy_fit = thz_hybrid_fit{0.80f, 0.20f, s};
        P.x0_spline = s;
        P.y0_spline = s;
        for (float fc : {0.01f, 0.1f, 0.5f, 1.7f, 9.0f, 50.0f}) {
            (void)hybrid_eval(P.wx_fit, fc);
            if (nk) { (void)spline_eval(s, fc); (void)spline_eval_const(s, fc); }
            for (float dd : {0.05f, 0.5f, 3.0f})
                for (int img : {16, 33, 128}) {
                    const BandPsf bp = band_psf(P, fc, dd, dd, img, img + 5);
                    if (bp.rows * bp.cols != (int)bp.v.size()) { std::printf("band_psf size mismatch\n"); return 1; }
                }
        }
    }
    for (size_t nt : {2, 500, 1001}) {
        const auto t = axis(nt, 0.0f, 0.05f);
        for (int nb : {2, 3, 25, 60}) {
            std::vector<float> filt, cen;
            filter_bank(nb, 0.1, 10.0, 0.5, t.data(), filt, cen);
            filter_bank(nb, 0.25, 4.0, 0.01, t.data(), filt, cen);
        }
    }
    std::printf("host math done\n");
    return 0;
}
