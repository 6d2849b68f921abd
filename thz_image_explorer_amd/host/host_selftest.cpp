// host_selftest.cpp — the reference's on-path unit tests re-created against the
// C++ host mirror (so they read like the originals), plus a default-chain run on
// a cube handed over by the Python test for comparison with the oracle.
//   usage: host_selftest <dir>     reads <dir>/cube.bin, writes <dir>/out.bin
#include "thz_host.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <mutex>
#include <optional>
#include <string>
#include <thread>

using namespace thzhost;

static int g_fail = 0;
#define CHECK(cond, name)                                             \
    do {                                                              \
        if (!(cond)) { std::printf("FAIL %s: %s\n", name, #cond); ++g_fail; } \
    } while (0)

static std::vector<float> linspace(float a, float b, size_t n)
{
    std::vector<float> v(n);
    const float step = n > 1 ? (b - a) / (float)(n - 1) : 0.0f;
    for (size_t i = 0; i < n; ++i) v[i] = a + step * (float)i;
    return v;
}

static ScannedImageFilterData make_input(const std::vector<float> &data, size_t w, size_t h,
                                         const std::vector<float> &time)
{
    ScannedImageFilterData in;
    in.width = w; in.height = h;
    in.time = time;
    in.data.upload(data.data(), data.size());
    in.has_plan = true;
    in.frequency.resize(time.size() / 2 + 1);
    for (size_t i = 0; i < in.frequency.size(); ++i) in.frequency[i] = (float)i / 50.0f;
    in.phases.resize(w * h * in.nf()); in.phases.zero();
    in.amplitudes.resize(w * h * in.nf()); in.amplitudes.zero();
    in.fft.resize(w * h * in.nf() * 2); in.fft.zero();
    in.img.resize(w * h); in.img.zero();
    return in;
}

// math_tools.rs:843-897
static void test_fft_roundtrip()
{
    const size_t n = 128;
    std::vector<float> data(n);
    const float pi = 3.14159274f;
    for (size_t t = 0; t < n; ++t) {
        const float tt = (float)t / (float)n;
        data[t] = std::sin(2.0f * pi * 3.0f * tt) + 0.5f * std::cos(2.0f * pi * 7.0f * tt);
    }
    ScannedImageFilterData input = make_input(data, 1, 1, linspace(0.0f, 1.0f, n));
    ConfigContainer config;
    config.fft_window_type = FftWindowType::AdaptedBlackman;
    config.fft_window = {0.0f, 0.0f};
    config.avg_in_fourier_space = false;
    const ScannedImageFilterData after_fft = math_tools::fft(input, config);
    const std::vector<float> expected_time = after_fft.data.download();
    const ScannedImageFilterData after_ifft = math_tools::ifft(after_fft, config);
    const std::vector<float> got = after_ifft.data.download();
    bool ok = true;
    for (size_t t = 0; t < n; ++t) ok = ok && std::fabs(got[t] - expected_time[t]) <= 1e-4f;
    CHECK(ok, "test_fft_roundtrip");
    CHECK(after_ifft.avg_signal_fft.size() == n / 2 + 1, "test_fft_roundtrip(avg)");
    std::printf("PASS? test_fft_roundtrip done\n");
}

// band_pass_fd.rs:475-567
static void test_fd_bandpass()
{
    const size_t n = 256, k = 9;
    std::vector<float> data(n);
    const float pi = 3.14159274f;
    for (size_t t = 0; t < n; ++t) data[t] = std::sin(2.0f * pi * (float)k * (float)t / (float)n);
    ScannedImageFilterData input = make_input(data, 1, 1, linspace(0.0f, 1.0f, n));
    ConfigContainer config;
    config.fft_window = {0.0f, 0.0f};
    const ScannedImageFilterData spec = math_tools::fft(input, config);
    FrequencyDomainBandPass f;
    f.low = spec.frequency[k - 2];
    f.high = spec.frequency[k + 2];
    f.window_width = 0.0;
    GuiSettingsContainer gui;
    ProgressLock pl = std::make_shared<std::pair<std::mutex, std::optional<float>>>();
    std::atomic<bool> abort{false};
    const ScannedImageFilterData out = f.filter(spec, gui, pl, abort);
    CHECK(out.fft.size() == spec.fft.size() && out.amplitudes.size() == spec.amplitudes.size(), "fd shapes");
    const std::vector<float> amp = out.amplitudes.download();
    bool zeros = true;
    float inside = 0.0f;
    for (size_t i = 0; i < amp.size(); ++i) {
        if (i < k - 2 || i > k + 2) zeros = zeros && amp[i] == 0.0f;
        else inside += amp[i];
    }
    CHECK(zeros, "fd exact zeros outside the band");
    CHECK(inside > 0.0f, "fd energy inside the band");
    std::printf("PASS? test_fd_bandpass done\n");
}

// band_pass_td_before_fft.rs:390-443
static void test_td_bandpass()
{
    const size_t n = 256;
    const std::vector<float> time = linspace(0.0f, 1.0f, n);
    std::vector<float> data(n);
    for (size_t t = 0; t < n; ++t) data[t] = std::sin(2.0f * 3.14159274f * 5.0f * time[t]);
    ScannedImageFilterData input = make_input(data, 1, 1, time);
    TimeDomainBandPassBeforeFFT f;
    f.low = 0.25; f.high = 0.55; f.window_width = 0.0;
    GuiSettingsContainer gui;
    ProgressLock pl = std::make_shared<std::pair<std::mutex, std::optional<float>>>();
    std::atomic<bool> abort{false};
    const ScannedImageFilterData out = f.filter(input, gui, pl, abort);
    const std::vector<float> o = out.data.download();
    size_t lower = 0, upper = n - 1;
    for (size_t i = 0; i < n; ++i) if (time[i] >= 0.25f) { lower = i; break; }
    for (size_t i = 0; i < n; ++i) if (time[i] >= 0.55f) { upper = i; break; }
    bool zeros = true;
    float inside = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        if (i < lower || i >= upper) zeros = zeros && o[i] == 0.0f;
        else inside += std::fabs(o[i]);
    }
    CHECK(o.size() == n, "td shape");
    CHECK(zeros, "td exact zeros outside the window");
    CHECK(inside > 0.0f, "td energy inside the window");
    std::printf("PASS? test_td_bandpass done\n");
}

// tilt_compensation.rs:303-389
static void test_tilt()
{
    const size_t n = 64, impulse = 10;
    const std::vector<float> time = linspace(0.0f, 0.05f * ((float)n - 1.0f), n);
    std::vector<float> data(2 * 2 * n, 0.0f);
    data[(1 * 2 + 1) * n + impulse] = 1.0f;
    ScannedImageFilterData input = make_input(data, 2, 2, time);
    input.dx = 1.0f; input.dy = 1.0f;
    GuiSettingsContainer gui;
    ProgressLock pl = std::make_shared<std::pair<std::mutex, std::optional<float>>>();
    std::atomic<bool> abort{false};
    for (double tx : {10.0, 0.0}) {
        TiltCompensation f;
        f.tilt_x = tx;
        const ScannedImageFilterData out = f.filter(input, gui, pl, abort);
        const size_t steps = thz_host_tilt_plan(time.data(), n, 2, 2, tx, 0.0, 1.0f, 1.0f, nullptr, nullptr);
        CHECK(out.time.size() == n + 2 * steps, "tilt extended length");
        CHECK((tx != 0.0) == (steps > 0), "tilt extension only with tilt");
        const std::vector<float> tr = out.data.download((1 * 2 + 1) * out.nt(), out.nt());
        size_t peak = 0;
        for (size_t i = 0; i < tr.size(); ++i) if (tr[i] > tr[peak]) peak = i;
        CHECK(peak == impulse + steps, "tilt centre-pixel impulse index");
    }
    std::printf("PASS? test_tilt done\n");
}

// deconvolution.rs:1139-1177: 2x2 image -> guard -> shape preserved, data unchanged
static void test_deconvolution_small()
{
    const size_t n = 64;
    std::vector<float> data(2 * 2 * n, 0.0f);
    data[5] = 1.0f;
    ScannedImageFilterData input = make_input(data, 2, 2, linspace(0.0f, 3.15f, n));
    input.dx = 1.0f; input.dy = 1.0f;
    GuiSettingsContainer gui;  // empty PSF, like a fresh GuiSettingsContainer::new()
    ProgressLock pl = std::make_shared<std::pair<std::mutex, std::optional<float>>>();
    std::atomic<bool> abort{false};
    Deconvolution f;
    const ScannedImageFilterData out = f.filter(input, gui, pl, abort);
    CHECK(out.data.size() == input.data.size(), "deconvolution shape");
    CHECK(out.data.download() == data, "deconvolution guard returns the input");
    std::printf("PASS? test_deconvolution_small done\n");
}

static void run_default_chain(const std::string &dir)
{
    std::FILE *fi = std::fopen((dir + "/cube.bin").c_str(), "rb");
    if (!fi) { std::printf("SKIP chain: no cube.bin\n"); return; }
    int32_t dims[3];
    float dxy[2];
    if (std::fread(dims, 4, 3, fi) != 3 || std::fread(dxy, 4, 2, fi) != 2) { std::fclose(fi); ++g_fail; return; }
    const size_t nx = dims[0], ny = dims[1], nt = dims[2];
    std::vector<float> time(nt), cube(nx * ny * nt);
    if (std::fread(time.data(), 4, nt, fi) != nt || std::fread(cube.data(), 4, cube.size(), fi) != cube.size()) {
        std::fclose(fi); ++g_fail; return;
    }
    std::fclose(fi);
    Pipeline pipe;
    // order of main.rs:194-247
    const char *expect[] = {"initial", "scaling", "Tilt Compensation", "Time Band Pass", "fft",
                            "Frequency Band Pass", "ifft", "Time Band Pass", "Deconvolution"};
    CHECK(pipe.filter_chain.size() == 9, "chain length");
    for (size_t i = 0; i < pipe.filter_chain.size() && i < 9; ++i) {
        const std::string &id = pipe.filter_chain[i];
        std::string name = id;
        for (auto &f : FilterRegistry::global().filters)
            if (f.first == id) name = f.second->config().name;
        CHECK(name == expect[i], "chain order");
    }
    pipe.open(ScannedImageFilterData::from_host_cube(cube.data(), nx, ny, time, dxy[0], dxy[1]));
    pipe.filter_data[0].rois["roi-1"] = {"pentagon", Polygon{{1, 1}, {5, 1}, {6, 4}, {3, 6}, {1, 4}}};
    pipe.update_filter(1);  // "Calculate All Filters"
    if (std::getenv("THZ_SELFTEST_VERBOSE")) {
        for (size_t i = 0; i < pipe.filter_data.size(); ++i) {
            const auto &d = pipe.filter_data[i];
            float mx = 0.0f, mf = 0.0f;
            for (float v : d.data.download()) mx = std::fmax(mx, std::fabs(v));
            for (float v : d.fft.download()) mf = std::fmax(mf, std::fabs(v));
            std::printf("stage %zu %-40s nt=%zu max|data|=%g max|fft|=%g\n", i, pipe.filter_chain[i].c_str(), d.nt(), mx, mf);
        }
    }
    const ScannedImageFilterData &spec = pipe.filter_data[pipe.filter_uuid_to_index["ifft"]];
    const ScannedImageFilterData &last = pipe.filter_data.back();
    std::FILE *fo = std::fopen((dir + "/out.bin").c_str(), "wb");
    auto put = [&](const std::vector<float> &v) { std::fwrite(v.data(), 4, v.size(), fo); };
    put(spec.fft.download());
    put(spec.amplitudes.download());
    put(spec.phases.download());
    put(last.data.download());
    put(last.img.download());
    put(spec.avg_signal_fft);
    put(spec.avg_phase_fft);
    put(spec.roi_data.at("roi-1").second);
    std::fclose(fo);
    CHECK(last.nt() == nt && last.data.size() == cube.size(), "chain output shape");
    // partial recompute from the fft stage (SetFFTWindow* -> Filter(fft_index), data_thread.rs:813-836)
    pipe.config.fft_window = {0.5f, 3.0f};
    pipe.update_filter(pipe.fft_index + 1);
    CHECK(pipe.filter_data.back().data.size() == cube.size(), "partial recompute shape");
    // second run: down-scaling by 2 and averaging in Fourier space
    // (SetDownScaling -> Filter(1), SetAvgInFourierSpace -> Filter(fft_index))
    pipe.config = ConfigContainer();
    pipe.config.scale_factor = 2;
    pipe.config.avg_in_fourier_space = true;
    pipe.update_filter(1);
    {
        const ScannedImageFilterData &sp = pipe.filter_data[pipe.filter_uuid_to_index["ifft"]];
        const ScannedImageFilterData &la = pipe.filter_data.back();
        CHECK(sp.width == nx / 2 && sp.height == ny / 2 && sp.scaling == 2, "scaled container shape");
        CHECK(la.img.size() == nx * ny, "scaled image is expanded back to the original size");
        std::FILE *f2 = std::fopen((dir + "/out2.bin").c_str(), "wb");
        auto put2 = [&](const std::vector<float> &v) { std::fwrite(v.data(), 4, v.size(), f2); };
        put2(la.data.download());
        put2(la.img.download());
        put2(sp.avg_data);
        put2(sp.avg_signal_fft);
        put2(sp.roi_data.at("roi-1").second);
        std::fclose(f2);
    }
    std::printf("PASS? default chain done (%zux%zux%zu)\n", nx, ny, nt);
}

// OpenFile on a .thzimg (io.rs:496-631 through libthzio.so): streamed in 3-row slabs, then
// the default chain; <dir>/scan.thzimg in, <dir>/out3.bin out (raw cube after the bias
// subtraction, image of the loader, final cube)
static void run_file_chain(const std::string &dir)
{
    const std::string path = dir + "/scan.thzimg";
    std::FILE *probe = std::fopen(path.c_str(), "rb");
    if (!probe) { std::printf("SKIP file chain: no scan.thzimg\n"); return; }
    std::fclose(probe);
    std::map<std::string, std::string> md;
    ScannedImageFilterData scan = io::open_scan_from_thz(path, &md, 3);
    CHECK(md.count("width") == 1 && md.count("dx [mm]") == 1, "metadata map");
    CHECK(scan.dx && *scan.dx == 0.5f && scan.dy && *scan.dy == 0.25f && !scan.x_min, "geometry from metadata");
    std::FILE *fo = std::fopen((dir + "/out3.bin").c_str(), "wb");
    auto put = [&](const std::vector<float> &v) { std::fwrite(v.data(), 4, v.size(), fo); };
    put(scan.data.download());
    put(scan.img.download());
    Pipeline pipe;
    pipe.open(std::move(scan));
    pipe.update_filter(1);
    put(pipe.filter_data.back().data.download());
    std::fclose(fo);
    const auto pulse = io::open_pulse_from_thz(path);  // a scan file is not a pulse file
    CHECK(pulse.first.empty() && pulse.second.empty(), "open_pulse_from_thz on a scan file");
    bool threw = false;
    try { io::open_scan_from_thz(dir + "/missing.thz"); } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw, "missing file throws");
}

// Two "devices" on one GPU: a thz_group whose two members are two contexts on device 0 —
// the same-device form of the group, collectives as device-local copies — against one session over the whole
// cube: the image must be the same bit for bit, the pixel means to rounding.
static void test_group_two_members()
{
    const size_t nx = 10, ny = 6, nt = 1024;
    std::vector<float> time = linspace(1000.0f, 1000.0f + 0.05f * (float)(nt - 1), nt), cube(nx * ny * nt);
    for (size_t p = 0; p < nx * ny; ++p)
        for (size_t t = 0; t < nt; ++t) {
            const float z = ((float)t * 0.05f - 10.0f - 0.01f * (float)p) / 0.35f;
            cube[p * nt + t] = (1.0f + 0.01f * (float)(p % 7)) * (-z * std::exp(-z * z));
        }
    thz_chain_cfg cfg;
    CHECK(thz_chain_cfg_default(time.data(), nt, &cfg) == THZ_OK, "chain defaults");
    // one session over everything
    thz_ctx *ctx = nullptr;
    thz_session *one = nullptr;
    CHECK(thz_create(0, &ctx) == THZ_OK, "thz_create");
    CHECK(thz_session_create(ctx, nx, ny, nt, time.data(), 0.5f, 0.5f, &one) == THZ_OK, "session");
    CHECK(thz_session_upload(one, cube.data(), 0) == THZ_OK, "upload");
    CHECK(thz_session_recompute(one, &cfg) == THZ_OK, "recompute");
    std::vector<float> img1(nx * ny), avg1(nt / 2 + 1), img2(nx * ny), avg2(nt / 2 + 1);
    CHECK(thz_session_download(one, THZ_BUF_IMG, 0, nx * ny, img1.data()) == THZ_OK, "image");
    CHECK(thz_session_download(one, THZ_BUF_AVG_AMPLITUDES, 0, 1, avg1.data()) == THZ_OK, "means");
    thz_session_destroy(one);
    thz_destroy(ctx);
    // two members on device 0
    const int devs[2] = {0, 0};
    thz_group *g = nullptr;
    thz_group_session *gs = nullptr;
    CHECK(thz_group_create(devs, 2, &g) == THZ_OK, "thz_group_create");
    CHECK(thz_group_world(g) == 2 && thz_group_local_count(g) == 2, "two members");
    size_t x0 = 0, n = 0;
    CHECK(thz_host_slab(nx, 2, 1, &x0, &n) == THZ_OK && x0 == 5 && n == 5, "slab rule");
    CHECK(thz_group_session_create(g, nx, ny, nt, time.data(), 0.5f, 0.5f, &gs) == THZ_OK, "group session");
    CHECK(thz_group_session_upload(gs, cube.data(), 0) == THZ_OK, "group upload");
    CHECK(thz_group_session_recompute(gs, &cfg, 1, THZ_GATHER_SMALL) == THZ_OK, "group recompute");
    CHECK(thz_group_session_download(gs, THZ_BUF_IMG, 0, nx * ny, img2.data()) == THZ_OK, "gathered image");
    CHECK(thz_group_session_download(gs, THZ_BUF_AVG_AMPLITUDES, 0, 1, avg2.data()) == THZ_OK, "group means");
    CHECK(img1 == img2, "slabs reproduce the whole-cube image bit for bit");
    float worst = 0.0f, scale = 0.0f;
    for (size_t k = 0; k < avg1.size(); ++k) {
        worst = std::fmax(worst, std::fabs(avg1[k] - avg2[k]));
        scale = std::fmax(scale, std::fabs(avg1[k]));
    }
    CHECK(worst <= 2e-6f * scale, "all-reduced means");
    float dummy = 0.0f;
    CHECK(thz_group_session_download(gs, THZ_BUF_DATA, 0, 1, &dummy) == THZ_ERR_NOT_READY, "ungathered buffer is absent");
    thz_group_session_destroy(gs);
    thz_group_destroy(g);
}

// filter.rs:631-637 / cancellable_loops: a click on the abort button DURING the call ends it
static void test_deconvolution_live_abort(const std::string &dir)
{
    // psf.npz is not readable from C++ without an npz reader: the Python side hands the splines over as psf.bin
    std::FILE *f = std::fopen((dir + "/psf.bin").c_str(), "rb");
    if (!f) { std::printf("SKIP test_deconvolution_live_abort (no psf.bin)\n"); return; }
    auto rd = [&](std::vector<float> &v) {
        int32_t n = 0;
        if (std::fread(&n, 4, 1, f) != 1) n = 0;
        v.resize((size_t)n);
        if (n && std::fread(v.data(), 4, (size_t)n, f) != (size_t)n) v.clear();
    };
    GuiSettingsContainer gui;
    float base[4] = {0, 0, 0, 0};
    if (std::fread(base, 4, 4, f) != 4) { std::fclose(f); return; }
    gui.psf.wx_a = base[0]; gui.psf.wx_b = base[1]; gui.psf.wy_a = base[2]; gui.psf.wy_b = base[3];
    for (int i = 0; i < 4; ++i) { rd(gui.psf.k[i]); rd(gui.psf.v[i]); rd(gui.psf.a[i]); rd(gui.psf.b[i]); rd(gui.psf.c[i]); rd(gui.psf.d[i]); }
    std::fclose(f);
    const size_t w = 64, h = 64, nt = 256;
    std::vector<float> time = linspace(1000.0f, 1000.0f + 0.05f * (float)(nt - 1), nt), data(w * h * nt);
    for (size_t p = 0; p < w * h; ++p)
        for (size_t t = 0; t < nt; ++t) {
            const float z = ((float)t * 0.05f - 5.0f) / 0.35f;
            data[p * nt + t] = (0.4f + 0.6f * (float)((p / 3) % 2)) * (-z * std::exp(-z * z));
        }
    ScannedImageFilterData input = make_input(data, w, h, time);
    input.dx = 0.5f; input.dy = 0.5f;
    Deconvolution flt;
    flt.n_iterations = 5000; flt.n_filters = 25; flt.start_freq = 0.1f;  // long enough (>= 100 ms) for a click to land inside the call
    flt.end_freq = 10.0f; flt.win_width = 0.5f;
    ProgressLock pl = std::make_shared<std::pair<std::mutex, std::optional<float>>>();
    std::atomic<bool> abort{false};
    float seen = -1.0f;
    std::thread clicker([&] {   // the GUI thread: watches the progress bar, clicks the abort button once it moves
        for (int i = 0; i < 4000 && !abort.load(); ++i) {
            {
                std::lock_guard<std::mutex> g(pl->first);
                if (pl->second && *pl->second > 0.0f) { seen = *pl->second; abort.store(true); }
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        abort.store(true);
    });
    const ScannedImageFilterData out = flt.filter(input, gui, pl, abort);
    clicker.join();
    CHECK(seen > 0.0f && seen < 1.0f, "progress was published while the call ran");
    CHECK(out.data.download() == data, "an abort during the call returns the input");
    {
        std::lock_guard<std::mutex> g(pl->first);
        CHECK(!pl->second.has_value(), "progress cleared after the call");
    }
}

int main(int argc, char **argv)
{
    try {
        test_group_two_members();
        if (argc > 1) test_deconvolution_live_abort(argv[1]);
        test_fft_roundtrip();
        test_fd_bandpass();
        test_td_bandpass();
        test_tilt();
        test_deconvolution_small();
        if (argc > 1) run_default_chain(argv[1]);
        if (argc > 1) run_file_chain(argv[1]);
    } catch (const std::exception &e) {
        std::printf("FAIL exception: %s\n", e.what());
        return 2;
    }
    std::printf(g_fail ? "SELFTEST FAILED (%d)\n" : "SELFTEST OK (%d failures)\n", g_fail);
    return g_fail ? 1 : 0;
}
