// tsan_driver.cpp — runs the emulated kernels (every lane a host thread) under ThreadSanitizer.
// TEST INFRASTRUCTURE ONLY.  A data race between two host threads here is an LDS / global-memory access
// pair of two lanes that no barrier (__syncthreads, wave_sync) orders — code that only works while a wave
// runs in lock-step.  Build and run: tests/emu/run_tsan.sh
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

extern "C" {
int emu_fft_fwd(int nt, size_t npix, const float *in, const float *wa, const float *wb, float *data_out, float *fft,
                float *amp, float *ph, const float *mask);
int emu_fft_inv(int nt, size_t npix, const float *fft, const float *win, float *out, float *img);
int emu_pipeline(int nt, size_t npix, const float *raw, const float *pre, const float *mask, const float *post,
                 float *fft, float *amp, float *ph, float *out, float *img);
int emu_pipeline_sums(int nt, size_t npix, const float *raw, const float *pre, const float *mask, const float *cmask,
                      const float *post, float *fft, float *amp, float *ph, float *out, float *img, float *sums);
void emu_allow_f(int on);
void emu_set_grid_cap(int blocks);
int emu_rl_iteration(int h, int w, int pr, int pc, int mode, const float *psf, const float *d, const float *u,
                     int tiled, float *t_out, float *u_out);
int emu_rl_iteration_sep(int h, int w, int pr, int pc, int mode, const float *psf, const float *fx, const float *fy,
                         const float *d, const float *u, int tiled, float *t_out, float *u_out);
int emu_dc_chain(int M, int nt, size_t npix, int n_bands, int shift, const float *in, const float *H,
                 const float *gain, int use_f, float *energy, float *out, float *img);
int emu_dc_energy_pv(int M, int nt, size_t npix, int n_bands, int n_taps, const float *in, const float *filters,
                     float *energy);
int emu_td_window(size_t npix, int nt, const float *in, const float *win, float *out);
int emu_pixel_sum(size_t nrows, size_t L, const float *arr, float *out);
int emu_gather_sum(const float *arr, size_t len, const uint32_t *list, uint32_t count, float div, float *out);
int emu_scale3d(const float *arr, size_t nx, size_t ny, size_t L, size_t s, float *out);
int emu_tilt(size_t npix, int nt_in, int nt_out, const float *in, const float *taper, const int *ins, float *out);
int emu_voxel_opacity(size_t npix, int nt, const float *data, const float *kernel, int radius, float contrast,
                      float opacity_threshold, float *out);
int emu_select_hist(const float *vals, size_t n, int level, uint32_t prefix, unsigned long long *hist);
}

static std::vector<float> noise(size_t n, unsigned seed, float lo = -1.0f, float hi = 1.0f)
{
    std::vector<float> v(n);
    uint32_t s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        v[i] = lo + (hi - lo) * (float)(s >> 8) / 16777216.0f;
    }
    return v;
}

static void chain(int nt, size_t npix, int allow_f)
{
    emu_allow_f(allow_f);
    const size_t nf = (size_t)nt / 2 + 1;
    auto x = noise(npix * nt, (unsigned)nt);
    auto pre = noise((size_t)nt, 1, 0.5f, 1.0f), post = noise((size_t)nt, 2, 0.5f, 1.0f), mask = noise(nf, 3, 0.0f, 1.0f);
    std::vector<float> fft(npix * nf * 2), amp(npix * nf), ph(npix * nf), out(npix * nt), img(npix), dat(npix * nt);
    int rc = emu_pipeline(nt, npix, x.data(), pre.data(), mask.data(), post.data(), fft.data(), amp.data(), ph.data(),
                          out.data(), img.data());
    if (rc == -2) {  // no fused kernel for this length: the staged pair
        emu_fft_fwd(nt, npix, x.data(), pre.data(), nullptr, dat.data(), fft.data(), amp.data(), ph.data(), mask.data());
        emu_fft_inv(nt, npix, fft.data(), post.data(), out.data(), img.data());
    } else {
        emu_fft_fwd(nt, npix, x.data(), pre.data(), nullptr, dat.data(), fft.data(), amp.data(), ph.data(), mask.data());
        emu_fft_inv(nt, npix, fft.data(), post.data(), out.data(), img.data());
    }
    std::printf("chain nt=%d allow_f=%d done\n", nt, allow_f);
    std::fflush(stdout);
}

int main(int argc, char **argv)
{
    // arguments: the kernel families to run (none: all of them)
    auto want = [&](const char *name) {
        if (argc <= 1) return true;
        for (int i = 1; i < argc; ++i)
            if (std::string(argv[i]) == name) return true;
        return false;
    };
    if (want("f")) { chain(1024, 11, 1); chain(2048, 9, 1); chain(4096, 9, 1); }
    if (want("sums")) {  // the fused chain with its pixel sums taken inside the launch: ticket-ordered LDS accumulation
        emu_allow_f(1);   // (FSums / PSums) — lock-free hand-over between the waves of a block, the case for this tool
        emu_set_grid_cap(1);  // one block: its waves go through several rounds, the last one ragged
        for (int nt : {1024, 4096, 1001}) {
            for (int cm = 0; cm < 2; ++cm) {
                const size_t npix = nt == 4096 ? 17 : (nt == 1001 ? 75 : 21), nf = (size_t)nt / 2 + 1;
                auto x = noise(npix * nt, (unsigned)nt + 77);
                auto pre = noise((size_t)nt, 1, 0.5f, 1.0f), post = noise((size_t)nt, 2, 0.5f, 1.0f), mask = noise(nf, 3, 0.0f, 1.0f);
                auto H = noise(2 * nf, 4, -1.0f, 1.0f);
                std::vector<float> fft(npix * nf * 2), amp(npix * nf), ph(npix * nf), out(npix * nt), img(npix), sums(2 * nf);
                const int rows = emu_pipeline_sums(nt, npix, x.data(), pre.data(), mask.data(), cm ? H.data() : nullptr, post.data(),
                                                   fft.data(), amp.data(), ph.data(), out.data(), img.data(), sums.data());
                std::printf("sums nt=%d cmask=%d rows=%d done\n", nt, cm, rows);
                std::fflush(stdout);
            }
        }
        emu_set_grid_cap(0);
    }
    if (want("g")) { chain(256, 7, 0); chain(1024, 5, 0); }
    if (want("fb")) { chain(1001, 5, 1); chain(300, 6, 1); }
    if (want("fbc")) { chain(1500, 3, 1); chain(3000, 3, 1); chain(5000, 3, 1); }
    if (want("rl")) {
        for (int mode = 0; mode < 2; ++mode) {
            const int h = 12, w = 13, pr = mode ? 19 : 7, pc = mode ? 21 : 9;
            const int H = h + 2 * (pr / 2), W = w + 2 * (pc / 2);
            auto d = noise((size_t)H * W, 5, 0.5f, 1.5f), u = noise((size_t)H * W, 6, 0.5f, 1.5f);
            auto psf = noise((size_t)pr * pc, 7, 0.0f, 1.0f);
            std::vector<float> t((size_t)H * W), un((size_t)H * W);
            emu_rl_iteration(h, w, pr, pc, mode, psf.data(), d.data(), u.data(), 1, t.data(), un.data());
            emu_rl_iteration(h, w, pr, pc, mode, psf.data(), d.data(), u.data(), 0, t.data(), un.data());
        }
        {   // six tiles of a narrow kernel: the second block has two groups of threads without a tile
            const int h = 20, w = 30, pr = 7, pc = 9;
            const int H = h + 2 * (pr / 2), W = w + 2 * (pc / 2);
            auto d = noise((size_t)H * W, 25, 0.5f, 1.5f), u = noise((size_t)H * W, 26, 0.5f, 1.5f);
            auto psf = noise((size_t)pr * pc, 27, 0.0f, 1.0f);
            std::vector<float> t((size_t)H * W), un((size_t)H * W);
            emu_rl_iteration(h, w, pr, pc, 0, psf.data(), d.data(), u.data(), 1, t.data(), un.data());
        }
        {   // a wide kernel given as an outer product: the two 1-D passes, halo rows not a multiple of 16
            const int h = 12, w = 22, pr = 19, pc = 21;
            const int H = h + 2 * (pr / 2), W = w + 2 * (pc / 2);
            auto d = noise((size_t)H * W, 35, 0.5f, 1.5f), u = noise((size_t)H * W, 36, 0.5f, 1.5f);
            auto fx = noise((size_t)pr, 37, 0.0f, 1.0f), fy = noise((size_t)pc, 38, 0.0f, 1.0f);
            std::vector<float> psf((size_t)pr * pc), t((size_t)H * W), un((size_t)H * W);
            for (int m = 0; m < pr; ++m)
                for (int n = 0; n < pc; ++n) psf[(size_t)m * pc + n] = fx[(size_t)m] * fy[(size_t)n];
            emu_rl_iteration_sep(h, w, pr, pc, 1, psf.data(), fx.data(), fy.data(), d.data(), u.data(), 2, t.data(), un.data());
        }
        std::printf("rl done\n");
    }
    if (want("dc")) {
        const int M = 1024, nt = 300, nb = 2;
        const size_t npix = 3, nk = M / 2 + 1;
        auto x = noise(npix * nt, 8), H = noise((size_t)nb * nk * 2, 9, -0.01f, 0.01f), g = noise((size_t)nb * npix, 10, 0.5f, 1.5f);
        std::vector<float> en((size_t)nb * npix), out(npix * nt), img(npix);
        for (int use_f = 0; use_f < 2; ++use_f)
            emu_dc_chain(M, nt, npix, nb, 249, x.data(), H.data(), g.data(), use_f, en.data(), out.data(), img.data());
        {   // the two-values-per-lane plans of the core (M = 2048, 4096)
            for (int M2 : {2048, 4096}) {
                const int nt2 = M2 - 600;
                const size_t nk2 = (size_t)M2 / 2 + 1;
                auto x2 = noise(npix * nt2, 18), H2 = noise((size_t)nb * nk2 * 2, 19, -0.01f, 0.01f);
                std::vector<float> out2(npix * nt2);
                emu_dc_chain(M2, nt2, npix, nb, 249, x2.data(), H2.data(), g.data(), 1, en.data(), out2.data(), img.data());
            }
        }
        {   // the band energies in Parseval form: the edges kernel's blocks walk the bands together and stage every band's
            // row in LDS one band ahead (block barriers, two halves of a table); more pixels than one round of waves, a
            // last round that is not full, a trace without edges (its transforms are skipped while the block goes on)
            const int taps = 499, nb3 = 3, M3 = 2048, nt3 = 1001;
            const size_t npix3 = 21;
            auto x3 = noise(npix3 * nt3, 28), h3 = noise((size_t)nb3 * taps, 29, -0.05f, 0.05f);
            for (int i = 0; i < nt3; ++i)
                if (i < 260 || i >= nt3 - 260) x3[(size_t)5 * nt3 + i] = 0.0f;
            std::vector<float> en3((size_t)nb3 * npix3);
            emu_set_grid_cap(2);
            emu_dc_energy_pv(M3, nt3, npix3, nb3, taps, x3.data(), h3.data(), en3.data());
            emu_set_grid_cap(0);
        }
        std::printf("dc done\n");
    }
    if (want("helpers")) {
        const size_t npix = 9;
        const int nt = 333;
        auto x = noise(npix * nt, 11), wv = noise((size_t)nt, 12);
        std::vector<float> o(npix * nt);
        emu_td_window(npix, nt, x.data(), wv.data(), o.data());
        {   // nt = 256 CH: the window chunks of a lane live in registers
            auto x2 = noise(npix * 1024, 31), w2 = noise((size_t)1024, 32);
            std::vector<float> o2(npix * 1024);
            emu_td_window(npix, 1024, x2.data(), w2.data(), o2.data());
        }
        auto a = noise((size_t)300 * 130, 13);
        std::vector<float> s(130);
        emu_pixel_sum(300, 130, a.data(), s.data());
        std::vector<uint32_t> lst = {5, 1, 200, 7, 7, 123, 64, 65, 66, 67};
        emu_gather_sum(a.data(), 130, lst.data(), (uint32_t)lst.size(), (float)lst.size(), s.data());
        auto c = noise((size_t)5 * 7 * 66, 14);
        std::vector<float> sc((size_t)2 * 3 * 66);
        emu_scale3d(c.data(), 5, 7, 66, 2, sc.data());
        {   // rows of whole 16-byte chunks: the vector path
            auto c2 = noise((size_t)5 * 7 * 132, 34);
            std::vector<float> sc2((size_t)2 * 3 * 132);
            emu_scale3d(c2.data(), 5, 7, 132, 2, sc2.data());
        }
        std::vector<int> ins = {0, 3, 10, 39, 40, 1, 2, 7, 20};
        std::vector<float> to(npix * 373);
        emu_tilt(npix, nt, 373, x.data(), wv.data(), ins.data(), to.data());
        std::printf("helpers done\n");
    }
    if (want("voxel")) {
        const size_t npix = 6;
        const int nt = 300, radius = 9;
        auto x = noise(npix * nt, 15);
        std::vector<float> k((size_t)2 * radius + 1, 1.0f / (2 * radius + 1)), o(npix * nt);
        emu_voxel_opacity(npix, nt, x.data(), k.data(), radius, 0.7f, 0.05f, o.data());
        std::vector<unsigned long long> hist(2048, 0);
        emu_select_hist(o.data(), o.size(), 0, 0, hist.data());
        std::printf("voxel done\n");
    }
    return 0;
}
