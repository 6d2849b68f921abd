// engine_selftest.cpp — drives the record-and-flush engine (thz_engine.hpp) through the patched stage walk the way
// the reference's data thread would: filters switched off and on again, one slider per chain position, the
// build-defined Frequency plugins, regions of interest, a scaled chain, and the Deconvolution stage on a group of
// two slabs.  Every scenario dumps the engine's results; tests/test_gpu_engine.py compares them with the oracle
// walking the same chain.  Checks that need no oracle (deferred show_data, group == single) are made here.
//
// usage: engine_selftest <dir>   (dir/cube.bin, dir/psf.bin as written by the test; outputs dir/<scenario>.bin)
#include "thz_engine.hpp"

#include <cmath>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

using namespace thzhost;

static int g_fail = 0;
#define CHECK(cond, msg)                                        \
    do {                                                        \
        if (!(cond)) {                                          \
            std::printf("FAIL: %s (%s:%d)\n", msg, __FILE__, __LINE__); \
            ++g_fail;                                           \
        }                                                       \
    } while (0)

struct Cube {
    int nx = 0, ny = 0, nt = 0;
    float dx = 1.0f, dy = 1.0f;
    std::vector<float> time, raw;
};

static Cube read_cube(const std::string &path)
{
    Cube c;
    std::ifstream f(path, std::ios::binary);
    int32_t dims[3] = {0, 0, 0};
    float d[2] = {1.0f, 1.0f};
    f.read(reinterpret_cast<char *>(dims), sizeof dims);
    f.read(reinterpret_cast<char *>(d), sizeof d);
    c.nx = dims[0]; c.ny = dims[1]; c.nt = dims[2]; c.dx = d[0]; c.dy = d[1];
    c.time.resize((size_t)c.nt);
    c.raw.resize((size_t)c.nx * c.ny * c.nt);
    f.read(reinterpret_cast<char *>(c.time.data()), (std::streamsize)(c.time.size() * sizeof(float)));
    f.read(reinterpret_cast<char *>(c.raw.data()), (std::streamsize)(c.raw.size() * sizeof(float)));
    return c;
}

static PsfArrays read_psf(const std::string &path)
{
    PsfArrays p;
    std::ifstream f(path, std::ios::binary);
    float base[4];
    f.read(reinterpret_cast<char *>(base), sizeof base);
    p.wx_a = base[0]; p.wx_b = base[1]; p.wy_a = base[2]; p.wy_b = base[3];
    for (int s = 0; s < 4; ++s)
        for (std::vector<float> *v : {&p.k[s], &p.v[s], &p.a[s], &p.b[s], &p.c[s], &p.d[s]}) {
            int32_t n = 0;
            f.read(reinterpret_cast<char *>(&n), sizeof n);
            v->resize((size_t)n);
            f.read(reinterpret_cast<char *>(v->data()), (std::streamsize)((size_t)n * sizeof(float)));
        }
    return p;
}

// [int32 nt_out, gx, gy, n_rois] [final cube gx*gy*nt_out] [img as the last container holds it] [avg amplitudes nf]
// per region with a polygon, in map order: [roi_signal_fft nf | roi_phase_fft nf | roi_data nt_out]
static void dump(GpuPipeline &p, const std::string &path)
{
    std::vector<float> cube;
    CHECK(p.eng.download_final(cube), "download_final");
    const ScannedImageFilterData &last = p.filter_data.back();
    const int32_t nto = (int32_t)p.eng.nt_out();
    const int32_t gx = (int32_t)last.width, gy = (int32_t)last.height;
    int32_t nroi = 0;
    for (auto &kv : last.roi_signal_fft) { (void)kv; ++nroi; }
    std::ofstream f(path, std::ios::binary);
    const int32_t head[4] = {nto, gx, gy, nroi};
    f.write(reinterpret_cast<const char *>(head), sizeof head);
    auto put = [&](const std::vector<float> &v) { f.write(reinterpret_cast<const char *>(v.data()), (std::streamsize)(v.size() * sizeof(float))); };
    CHECK(cube.size() == (size_t)gx * gy * nto, "final cube size");
    put(cube);
    put(last.img.download());
    put(last.avg_signal_fft);
    for (auto &kv : last.roi_signal_fft) {
        put(kv.second.second);
        put(last.roi_phase_fft.at(kv.first).second);
        put(last.roi_data.at(kv.first).second);
    }
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        std::printf("usage: engine_selftest <dir>\n");
        return 2;
    }
    const std::string dir = argv[1];
    const Cube c = read_cube(dir + "/cube.bin");
    GpuEngine eng({0});
    if (!eng.available()) {
        std::printf("FAIL: no GPU engine\n");
        return 1;
    }
    GpuPipeline p(eng);
    p.gui_settings.psf = read_psf(dir + "/psf.bin");
    p.open(c.raw.data(), (size_t)c.nx, (size_t)c.ny, c.time, c.dx, c.dy);
    p.filter_data[0].pixel_selected = {3, 2};
    const size_t i_tilt = p.index_of("Tilt Compensation"), i_tdb = p.index_of("Band-Pass Filter in Time Domain before the FFT."),
                 i_fd = p.index_of("Frequency Band Pass"), i_water = p.index_of("Water Line Notch"),
                 i_wiener = p.index_of("Reference Wiener Filter"), i_tda = p.index_of("Band-Pass Filter in Time Domain after the FFT."),
                 i_dec = p.index_of("Deconvolution");
    CHECK(i_tilt == 2 && i_tdb == 3 && p.filter_chain[4] == "fft" && i_fd == 5 && i_water == 6 && i_wiener == 7
              && p.filter_chain[8] == "ifft" && i_tda == 9 && i_dec == 10,
          "chain order: initial, scaling, Tilt, Time Band Pass, fft, Frequency plugins, ifft, Time Band Pass, Deconvolution");
    auto active = [&](size_t idx, bool on) { p.filters_active[p.filter_chain[idx]] = on; };

    // ---- 1: OpenFile -> Filter(1), defaults
    p.update_filter(1);
    dump(p, dir + "/default.bin");
    CHECK(eng.dirty_from() > 8, "flush leaves nothing pending");

    // ---- 2: each plugin switched off in turn (UpdateFilter(uuid) -> Filter(idx of the filter), data_thread.rs:907-921);
    //         the reference does not call filter() for an inactive plugin (:1185-1188)
    active(i_fd, false);
    p.update_filter(i_fd);
    CHECK(eng.pending().fd_active == 0, "an inactive Frequency Band Pass is off in the chain");
    dump(p, dir + "/fd_off.bin");
    active(i_fd, true);
    active(i_tdb, false);
    p.update_filter(i_tdb);
    CHECK(eng.pending().fd_active == 1 && eng.pending().td_before_active == 0, "fd on again, Time Band Pass off");
    dump(p, dir + "/tdb_off.bin");
    active(i_tdb, true);
    active(i_tda, false);
    p.update_filter(i_tdb);
    dump(p, dir + "/tda_off.bin");
    active(i_tda, true);
    active(i_tilt, false);
    p.update_filter(i_tilt);
    CHECK(eng.pending().tilt_active == 0 && eng.pending().td_after_active == 1, "tilt off, the others on");
    dump(p, dir + "/tilt_off.bin");
    active(i_tilt, true);
    p.update_filter(1);

    // ---- 3: one slider per chain position, the walk starting where the reference starts it
    auto *fd = dynamic_cast<FrequencyDomainBandPass *>(p.filter_by("Frequency Band Pass"));
    const std::vector<float> fd_plot_before = fd->signal_axis;
    fd->low = 0.4; fd->high = 2.5;
    p.update_filter(i_fd);
    {   // show_data ran AFTER the flush: the filter's plot is the selected pixel's band-passed spectrum of THIS walk
        std::vector<float> amp(eng.nt_out() / 2 + 1);
        thz_plot_out po{};
        po.filtered_signal_fft = amp.data();
        CHECK(eng.plot(3, 2, po), "plot");
        CHECK(fd->signal_axis == amp, "deferred show_data: the Frequency Band Pass plot shows the current results");
        CHECK(fd->signal_axis != fd_plot_before, "... and not the previous walk's");
        size_t zeros = 0;
        for (float v : amp) zeros += v == 0.0f;
        CHECK(zeros > amp.size() / 2, "0.4 - 2.5 THz leaves most bins exactly zero");
    }
    auto *tda = dynamic_cast<TimeDomainBandPassBeforeFFT *>(p.filter_by("Band-Pass Filter in Time Domain after the FFT."));
    tda->high = (double)c.time.back() - 4.0;
    p.update_filter(i_tda);   // chain position 7: only C2R + taper + image are re-run
    p.config.fft_window = {0.5f, 3.0f};
    p.update_filter(p.fft_index);   // SetFFTWindow*: Filter(fft_index), data_thread.rs:813-836
    dump(p, dir + "/sliders.bin");

    // ---- 4: the build-defined Frequency plugins: on, then off again (an inactive plugin must not linger)
    auto *water = dynamic_cast<WaterLineNotch *>(p.filter_by("Water Line Notch"));
    {
        std::ifstream lf(dir + "/water_lines.bin", std::ios::binary);
        int32_t n = 0;
        lf.read(reinterpret_cast<char *>(&n), sizeof n);
        water->lines_thz.resize((size_t)n);
        lf.read(reinterpret_cast<char *>(water->lines_thz.data()), (std::streamsize)((size_t)n * sizeof(float)));
        water->sigma_thz = 0.02f;
    }
    active(i_water, true);
    p.update_filter(i_water);
    dump(p, dir + "/water_on.bin");
    active(i_water, false);
    p.update_filter(i_water);
    dump(p, dir + "/water_off.bin");   // == sliders.bin

    // ---- 5: regions of interest + avg_in_fourier_space through the ifft stage
    p.filter_data[0].rois["roi-a"] = {"ROI 1", Polygon{{1, 1}, {5, 1}, {6, 4}, {3, 6}, {1, 4}}};
    p.filter_data[0].rois["roi-b"] = {"ROI 2", Polygon{{8, 2}, {14, 3}, {12, 12}}};
    p.filter_data[0].rois["roi-none"] = {"no polygon yet", std::nullopt};
    p.update_filter(1);
    dump(p, dir + "/rois.bin");
    p.config.avg_in_fourier_space = true;
    p.update_filter(p.fft_index);   // SetAvgInFourierSpace: Filter(fft_index)
    dump(p, dir + "/rois_fourier.bin");
    CHECK(p.filter_data.back().avg_data.size() == eng.nt_out(), "avg_data of the ifft stage");
    p.config.avg_in_fourier_space = false;

    // ---- 6: scaling (SetDownScaling -> Filter(1)); the image is expanded to the raw grid like data_thread.rs:1243-1285
    p.config.scale_factor = 2;
    p.update_filter(1);
    dump(p, dir + "/scaled.bin");
    CHECK(p.filter_data.back().img.size() == (size_t)(c.nx / 2 * 2) * (size_t)(c.ny / 2 * 2), "scaled image expanded s x s");
    p.config.scale_factor = 1;
    p.update_filter(1);

    // ---- 7: the Deconvolution stage, on one slab and on a group of two: the stage must see the WHOLE image
    auto small_bank = [](Filter *f) {   // a bank whose widest band PSF fits a 36 x 32 image (deconvolution.rs:873-885)
        auto *d = dynamic_cast<Deconvolution *>(f);
        d->n_iterations = 20; d->n_filters = 5; d->start_freq = 0.4f; d->end_freq = 3.0f; d->win_width = 0.5f;
    };
    small_bank(p.filter_by("Deconvolution"));
    active(i_dec, true);
    p.update_filter(i_dec);
    std::vector<float> dec_single;
    CHECK(eng.download_final(dec_single), "download_final");
    dump(p, dir + "/deconv_single.bin");
    {
        GpuEngine eng2({0, 0});
        CHECK(eng2.available(), "two slabs on device 0");
        GpuPipeline q(eng2);
        q.gui_settings.psf = p.gui_settings.psf;
        q.open(c.raw.data(), (size_t)c.nx, (size_t)c.ny, c.time, c.dx, c.dy);
        q.filter_data[0].pixel_selected = {3, 2};
        q.filter_data[0].rois = p.filter_data[0].rois;
        *dynamic_cast<FrequencyDomainBandPass *>(q.filter_by("Frequency Band Pass")) = *fd;
        dynamic_cast<TimeDomainBandPassBeforeFFT *>(q.filter_by("Band-Pass Filter in Time Domain after the FFT."))->high = tda->high;
        q.config = p.config;
        q.update_filter(1);
        // (reset() put the Time Band Pass bounds back to the full range: set the slider again, as the GUI would)
        dynamic_cast<TimeDomainBandPassBeforeFFT *>(q.filter_by("Band-Pass Filter in Time Domain after the FFT."))->high = tda->high;
        q.update_filter(q.index_of("Band-Pass Filter in Time Domain after the FFT."));
        std::vector<float> before;
        CHECK(eng2.download_final(before), "download_final");
        small_bank(q.filter_by("Deconvolution"));
        q.filters_active[q.filter_chain[q.index_of("Deconvolution")]] = true;
        q.update_filter(q.index_of("Deconvolution"));
        std::vector<float> dec_group;
        CHECK(eng2.download_final(dec_group), "download_final");
        CHECK(dec_group.size() == dec_single.size(), "group deconvolution: size");
        double err = 0.0, scale = 0.0, moved = 0.0;
        for (size_t i = 0; i < dec_group.size() && i < dec_single.size(); ++i) {
            err = std::fmax(err, std::fabs((double)dec_group[i] - dec_single[i]));
            scale = std::fmax(scale, std::fabs((double)dec_single[i]));
            moved = std::fmax(moved, std::fabs((double)dec_group[i] - before[i]));
        }
        std::printf("deconvolution, two slabs vs one: max |diff| %.3e of %.3e; the stage moved the cube by %.3e\n", err, scale, moved);
        CHECK(err <= 2e-6 * scale, "Deconvolution over two slabs == over one (the stage sees the whole image)");
        CHECK(moved > 1e-3 * scale, "... and it did run");
        dump(q, dir + "/deconv_group.bin");
        // a walk that starts elsewhere passes the stage's input through again (data_thread.rs:1139-1149)
        q.update_filter(q.index_of("Band-Pass Filter in Time Domain after the FFT."));
        std::vector<float> after;
        CHECK(eng2.download_final(after), "download_final");
        CHECK(after == before, "Deconvolution is a pass-through unless it is the filter being updated");
    }
    // switched off: the stage's input is the chain's output again
    active(i_dec, false);
    p.update_filter(i_dec);
    std::vector<float> undone;
    CHECK(eng.download_final(undone), "download_final");
    CHECK(undone != dec_single, "an inactive Deconvolution hands its input on");
    dump(p, dir + "/deconv_off.bin");

    std::printf(g_fail ? "ENGINE SELFTEST FAILED (%d)\n" : "ENGINE SELFTEST OK\n", g_fail);
    return g_fail ? 1 : 0;
}
