// thz_engine.cpp — see thz_engine.hpp.  rust/engine.rs, rust/math_tools_gpu.rs and rust/filters/*.rs are this
// file in the reference's language.
#include "thz_engine.hpp"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace thzhost {

static void log_error(const std::string &m) { std::fprintf(stderr, "[thzhost][error] %s\n", m.c_str()); }
static void log_warn(const std::string &m) { std::fprintf(stderr, "[thzhost][warn] %s\n", m.c_str()); }

int chain_position(FilterDomain d)
{
    switch (d) {
    case FilterDomain::TimeBeforeFFTPrioFirst: return kPosTilt;
    case FilterDomain::TimeBeforeFFT: return kPosTdBefore;
    case FilterDomain::Frequency: return kPosFrequency;
    case FilterDomain::TimeAfterFFT: return kPosTdAfter;
    case FilterDomain::TimeAfterFFTPrioLast: return kPosDeconvolution;
    }
    return kPosScaling;
}
int chain_position(const std::string &id)
{
    if (id == "scaling") return kPosScaling;
    if (id == "fft") return kPosFft;
    if (id == "ifft") return kPosIfft;
    return 0;
}

// ------------------------------------------------------------------ GpuEngine
GpuEngine::GpuEngine(const std::vector<int> &devices)
{
    if (thz_group_create(devices.data(), (int)devices.size(), &group_) != THZ_OK) {
        // no CPU compute path exists in the engine: the caller keeps the reference's own path
        log_error(std::string("thz_group_create failed (") + thz_group_last_error(nullptr) + "): GPU path disabled");
        group_ = nullptr;
    }
    std::memset(&pending_, 0, sizeof pending_);
    pending_.scale_factor = 1;
    pending_.want_means = 1;
}
GpuEngine::~GpuEngine()
{
    if (session_) thz_group_session_destroy(session_);
    if (group_) thz_group_destroy(group_);
}
GpuEngine &GpuEngine::instance()
{
    static GpuEngine e([] {
        std::vector<int> dev;
        if (const char *s = std::getenv("THZGPU_DEVICES")) {
            for (const char *p = s; *p;) {
                char *end = nullptr;
                const long v = std::strtol(p, &end, 10);
                if (end == p) break;
                dev.push_back((int)v);
                p = *end == ',' ? end + 1 : end;
            }
        }
        if (dev.empty()) dev.push_back(0);
        return dev;
    }());
    return e;
}
std::string GpuEngine::last_error() const { return group_ ? thz_group_last_error(group_) : "no GPU group"; }

bool GpuEngine::open_scan(const float *cube, size_t nx_, size_t ny_, const std::vector<float> &time, float dx, float dy,
                          bool subtract_bias)
{
    if (!group_) return false;
    if (session_) {
        thz_group_session_destroy(session_);
        session_ = nullptr;
    }
    if (thz_group_session_create(group_, nx_, ny_, time.size(), time.data(), dx, dy, &session_) != THZ_OK
        || thz_group_session_upload(session_, cube, subtract_bias ? 1 : 0) != THZ_OK) {
        log_error("open_scan: " + last_error());
        if (session_) thz_group_session_destroy(session_);
        session_ = nullptr;
        return false;
    }
    thz_chain_cfg_default(time.data(), time.size(), &pending_);
    nx = nx_; ny = ny_; nt = time.size();
    dirty_from_ = 1;
    fd_real_.clear();
    fd_cmask_.clear();
    plugins_dirty_ = true;
    rois_.clear();
    rois_dirty_ = false;
    return true;
}

void GpuEngine::begin_walk(int start_position) { touch(start_position); }
void GpuEngine::record_scaling(size_t s)
{
    pending_.scale_factor = (int32_t)s;
    touch(kPosScaling);
}
void GpuEngine::record_tilt(bool active, double tx, double ty)
{
    pending_.tilt_active = active ? 1 : 0;
    if (active) { pending_.tilt_x_deg = tx; pending_.tilt_y_deg = ty; }
    touch(kPosTilt);
}
void GpuEngine::record_td_before(bool active, double low, double high, double width)
{
    pending_.td_before_active = active ? 1 : 0;
    if (active) { pending_.td_before_low = low; pending_.td_before_high = high; pending_.td_before_width = width; }
    touch(kPosTdBefore);
}
void GpuEngine::record_fft(int window_type, float lower, float upper)
{
    pending_.fft_window = thz_window_cfg{window_type, lower, upper};
    touch(kPosFft);
}
void GpuEngine::record_fd(bool active, double low, double high, double width)
{
    pending_.fd_active = active ? 1 : 0;
    if (active) { pending_.fd_low = low; pending_.fd_high = high; pending_.fd_width = width; }
    touch(kPosFrequency);
}
void GpuEngine::record_water_lines(bool active, std::vector<float> mask)
{
    if (!active) mask.clear();
    if (mask != fd_real_) {  // (an unchanged multiplier does not invalidate the session's resident spectrum)
        fd_real_ = std::move(mask);
        plugins_dirty_ = true;
    }
    touch(kPosFrequency);
}
void GpuEngine::record_wiener(bool active, std::vector<float> cmask)
{
    if (!active) cmask.clear();
    if (cmask != fd_cmask_) {
        fd_cmask_ = std::move(cmask);
        plugins_dirty_ = true;
    }
    touch(kPosFrequency);
}
void GpuEngine::record_ifft(bool avg_in_fourier_space, const std::vector<std::pair<std::string, Polygon>> &rois)
{
    pending_.avg_in_fourier_space = avg_in_fourier_space ? 1 : 0;
    if (rois != rois_) {
        rois_ = rois;
        rois_dirty_ = true;
    }
    touch(kPosIfft);
}
void GpuEngine::record_td_after(bool active, double low, double high, double width)
{
    pending_.td_after_active = active ? 1 : 0;
    if (active) { pending_.td_after_low = low; pending_.td_after_high = high; pending_.td_after_width = width; }
    touch(kPosTdAfter);
}
void GpuEngine::note_inactive(const FilterConfig &cfg)
{
    switch (cfg.domain) {
    case FilterDomain::TimeBeforeFFTPrioFirst: record_tilt(false, 0.0, 0.0); break;
    case FilterDomain::TimeBeforeFFT: record_td_before(false, 0.0, 0.0, 0.0); break;
    case FilterDomain::Frequency:
        if (cfg.name == "Water Line Notch") record_water_lines(false, {});
        else if (cfg.name == "Reference Wiener Filter") record_wiener(false, {});
        else record_fd(false, 0.0, 0.0, 0.0);
        break;
    case FilterDomain::TimeAfterFFT: record_td_after(false, 0.0, 0.0, 0.0); break;
    case FilterDomain::TimeAfterFFTPrioLast:
        // the stage hands its input on: an earlier deconvolved cube must not stay the chain's output.  The tail of
        // the chain (C2R, Time Band Pass, image) is what restores the stage's input as the final cube.
        touch(kPosTdAfter);
        break;
    }
}

bool GpuEngine::flush()
{
    if (!session_) return false;
    const int n_local = thz_group_local_count(group_);
    if (plugins_dirty_) {
        for (int i = 0; i < n_local; ++i) {
            thz_session *s = thz_group_session_member(session_, i);
            const size_t nf = !fd_real_.empty() ? fd_real_.size() : fd_cmask_.size() / 2;
            if (thz_session_set_fd_filters(s, fd_real_.empty() ? nullptr : fd_real_.data(),
                                           fd_cmask_.empty() ? nullptr : fd_cmask_.data(), nf) != THZ_OK) {
                log_error("flush: set_fd_filters failed");
                return false;
            }
        }
        plugins_dirty_ = false;
    }
    if (rois_dirty_) {
        std::vector<size_t> counts;
        std::vector<uint64_t> flat;
        for (const auto &r : rois_) {
            counts.push_back(r.second.size());
            for (const auto &v : r.second) {
                flat.push_back((uint64_t)v.first);
                flat.push_back((uint64_t)v.second);
            }
        }
        if (thz_group_session_set_rois(session_, rois_.size(), counts.data(), flat.data()) != THZ_OK) {
            log_error("flush: set_rois: " + last_error());
            return false;
        }
        rois_dirty_ = false;
        touch(kPosIfft);
    }
    if (dirty_from_ > kPosDeconvolution) return true;  // nothing recorded since the last flush
    if (thz_group_session_recompute(session_, &pending_, dirty_from_, THZ_GATHER_SMALL) != THZ_OK) {
        log_error("flush: " + last_error());
        return false;
    }
    dirty_from_ = kPosDeconvolution + 1;
    return true;
}

int GpuEngine::deconvolve(const thz_psf &psf, const thz_deconv_cfg &cfg, ProgressLock &progress, const std::atomic<bool> &abort_flag)
{
    // everything in front of the stage must be on the device before it runs
    if (!flush()) return THZ_ERR_NOT_READY;
    // the engine polls a plain int between iteration batches and writes its progress into a float; a watcher
    // thread bridges them to the AtomicBool / RwLock<Option<f32>> of the reference's plugin interface
    volatile int abort_now = abort_flag.load(std::memory_order_relaxed) ? 1 : 0;
    volatile float prog = 0.0f;
    std::atomic<bool> done{false};
    std::thread watcher([&] {
        while (!done.load(std::memory_order_acquire)) {
            if (abort_flag.load(std::memory_order_relaxed)) abort_now = 1;
            if (progress) {
                std::lock_guard<std::mutex> g(progress->first);
                progress->second = prog;
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
    });
    const int rc = thz_group_session_deconvolve(session_, &psf, &cfg, &abort_now, const_cast<float *>(&prog));
    done.store(true, std::memory_order_release);
    watcher.join();
    if (progress) {
        std::lock_guard<std::mutex> g(progress->first);
        progress->second.reset();
    }
    return rc;
}

bool GpuEngine::image(std::vector<float> &img, size_t &gx, size_t &gy)
{
    if (!session_) return false;
    thz_session *s0 = thz_group_session_member(session_, 0);
    gx = nx; gy = ny;
    // the outputs' grid: every member's rows of it (a scaled recompute runs on one slab, the group's whole grid)
    size_t rows = 0, cols = 0;
    for (int i = 0; i < thz_group_local_count(group_); ++i) {
        size_t r = 0, c = 0;
        thz_session_grid(thz_group_session_member(session_, i), &r, &c, nullptr, nullptr);
        rows += r;
        cols = c;
    }
    if (thz_group_local_count(group_) == thz_group_world(group_)) { gx = rows; gy = cols; }
    img.assign(gx * gy, 0.0f);
    int rc = thz_group_session_download(session_, THZ_BUF_IMG, 0, gx * gy, img.data());
    if (rc == THZ_ERR_NOT_READY)  // before the first recompute: the upload's image of the raw grid (one slab)
        rc = thz_session_download(s0, THZ_BUF_IMG, 0, gx * gy, img.data());
    return rc == THZ_OK;
}

bool GpuEngine::averages(std::vector<std::complex<float>> &avg_fft, std::vector<float> &avg_amp, std::vector<float> &avg_phase)
{
    if (!session_) return false;
    const size_t nf = nt_out() / 2 + 1;
    avg_fft.assign(nf, {});
    avg_amp.assign(nf, 0.0f);
    avg_phase.assign(nf, 0.0f);
    return thz_group_session_download(session_, THZ_BUF_AVG_FFT, 0, 1, avg_fft.data()) == THZ_OK
           && thz_group_session_download(session_, THZ_BUF_AVG_AMPLITUDES, 0, 1, avg_amp.data()) == THZ_OK
           && thz_group_session_download(session_, THZ_BUF_AVG_PHASES, 0, 1, avg_phase.data()) == THZ_OK;
}

// the member whose slab holds raw row px (thz_host_slab: the partition the library itself uses)
thz_session *GpuEngine::owner_of(size_t px, size_t *local_px) const
{
    const int world = thz_group_world(group_);
    for (int i = 0; i < thz_group_local_count(group_); ++i) {
        size_t x0 = 0, n = 0;
        thz_host_slab(nx, world, thz_group_rank(group_, i), &x0, &n);
        if (px >= x0 && px < x0 + n) {
            *local_px = px - x0;
            return thz_group_session_member(session_, i);
        }
    }
    return nullptr;
}

bool GpuEngine::plot(size_t px, size_t py, const thz_plot_out &out)
{
    if (!session_) return false;
    size_t lx = 0;
    thz_session *s = owner_of(px, &lx);
    if (!s) return false;
    const size_t sf = pending_.scale_factor > 1 ? (size_t)pending_.scale_factor : 1;
    if (sf == 1 || thz_group_world(group_) == 1) return thz_session_plot(s, lx, py, &out) == THZ_OK;
    // Several slabs behind a scaling stage: the raw trace comes from the slab that holds row px, everything else from
    // the slab that holds the pixel's BLOCK — the one with the block's last raw row (thz_group_session_recompute)
    thz_plot_out raw{};
    raw.signal = out.signal;
    if (out.signal && thz_session_plot(s, lx, py, &raw) != THZ_OK) return false;
    thz_plot_out rest = out;
    rest.signal = nullptr;
    size_t lb = 0;
    thz_session *sb = owner_of((px / sf) * sf + sf - 1, &lb);
    return sb && thz_session_plot(sb, lb, py, &rest) == THZ_OK;
}

bool GpuEngine::roi(const std::string &uuid, const thz_roi_out &out)
{
    if (!session_) return false;
    for (size_t i = 0; i < rois_.size(); ++i)
        if (rois_[i].first == uuid) return thz_group_session_roi(session_, i, &out) == THZ_OK;
    return false;
}

size_t GpuEngine::nt_out() const { return session_ ? thz_session_nt_out(thz_group_session_member(session_, 0)) : 0; }
std::vector<float> GpuEngine::time_out() const
{
    std::vector<float> t(nt_out());
    if (session_ && !t.empty()) thz_session_time_out(thz_group_session_member(session_, 0), t.data());
    return t;
}
bool GpuEngine::download_final(std::vector<float> &cube)
{
    if (!session_) return false;
    cube.clear();
    const size_t nto = nt_out();
    for (int i = 0; i < thz_group_local_count(group_); ++i) {
        thz_session *s = thz_group_session_member(session_, i);
        size_t r = 0, c = 0;
        thz_session_grid(s, &r, &c, nullptr, nullptr);
        std::vector<float> part(r * c * nto);
        if (thz_session_download(s, THZ_BUF_DATA, 0, r * c, part.data()) != THZ_OK) return false;
        cube.insert(cube.end(), part.begin(), part.end());
    }
    return true;
}

// ------------------------------------------------------------------ math_tools on the engine
namespace math_tools_gpu {

ScannedImageFilterData shallow_clone(const ScannedImageFilterData &in)
{
    ScannedImageFilterData o;
    o.x_min = in.x_min; o.dx = in.dx; o.y_min = in.y_min; o.dy = in.dy;
    o.height = in.height; o.width = in.width; o.scaling = in.scaling; o.pixel_selected = in.pixel_selected;
    o.has_plan = in.has_plan; o.rois = in.rois; o.time = in.time; o.avg_data = in.avg_data; o.roi_data = in.roi_data;
    o.frequency = in.frequency; o.avg_fft = in.avg_fft; o.avg_signal_fft = in.avg_signal_fft; o.avg_phase_fft = in.avg_phase_fft;
    o.roi_signal_fft = in.roi_signal_fft; o.roi_phase_fft = in.roi_phase_fft;
    return o;  // img, data, fft, amplitudes, phases: empty — resident on the device
}

// chain position 1 (data_thread.rs:1109-1112); metadata exactly as math_tools.rs:250-270
ScannedImageFilterData scaling(GpuEngine &eng, const ScannedImageFilterData &input, const ConfigContainer &config)
{
    const size_t s = config.scale_factor;
    eng.record_scaling(s);
    ScannedImageFilterData output = shallow_clone(input);
    if (s > 1 && input.width / s > 0 && input.height / s > 0) {
        output.width = input.width / s;
        output.height = input.height / s;
        output.scaling = s;
        if (output.dx) output.dx = *output.dx * (float)s;
        if (output.dy) output.dy = *output.dy * (float)s;
        output.pixel_selected = {input.pixel_selected[0] / s, input.pixel_selected[1] / s};
    }
    return output;
}

// chain position 4 (data_thread.rs:1113-1116)
ScannedImageFilterData fft(GpuEngine &eng, const ScannedImageFilterData &input, const ConfigContainer &config)
{
    if (!input.has_plan) return input;  // math_tools.rs:332
    eng.record_fft((int)config.fft_window_type, config.fft_window[0], config.fft_window[1]);
    return shallow_clone(input);
}

// chain position 6 (data_thread.rs:1117-1120): the stage's regions and avg_in_fourier_space go with it
ScannedImageFilterData ifft(GpuEngine &eng, const ScannedImageFilterData &input, const ConfigContainer &config)
{
    std::vector<std::pair<std::string, Polygon>> rois;
    for (const auto &kv : input.rois)
        if (kv.second.second) rois.emplace_back(kv.first, *kv.second.second);  // (name, Some(polygon)), math_tools.rs:474-475
    eng.record_ifft(config.avg_in_fourier_space, rois);
    return shallow_clone(input);
}

bool finish_stage_walk(GpuEngine &eng, ScannedImageFilterData &last, const ConfigContainer &config)
{
    if (!eng.flush()) {
        log_error("gpu recompute failed (" + eng.last_error() + "): results of the previous run are kept");
        return false;
    }
    // the chain's time axis (a tilted cube's is longer) — the stages' containers carried it along already
    const size_t nto = eng.nt_out(), nf = nto / 2 + 1;
    // img: data_thread.rs:1242-1308 — on the scaled grid the value of a block fills its s x s pixels
    std::vector<float> img;
    size_t gx = 0, gy = 0;
    if (eng.image(img, gx, gy)) {
        const size_t s = last.scaling > 1 ? last.scaling : 1;
        if (s > 1) {
            std::vector<float> big(gx * s * gy * s, 0.0f);
            for (size_t x = 0; x < gx; ++x)
                for (size_t y = 0; y < gy; ++y)
                    for (size_t a = 0; a < s; ++a)
                        for (size_t b = 0; b < s; ++b) big[(x * s + a) * (gy * s) + (y * s + b)] = img[x * gy + y];
            last.img.upload(big.data(), big.size());
        } else {
            last.img.upload(img.data(), img.size());
        }
    }
    eng.averages(last.avg_fft, last.avg_signal_fft, last.avg_phase_fft);  // math_tools.rs:421-440
    // avg_data (math_tools.rs:442-470) rides in the plot copy-out's avg_signal when averaging in Fourier space
    if (config.avg_in_fourier_space) {
        last.avg_data.assign(nto, 0.0f);
        thz_plot_out po{};
        po.avg_signal = last.avg_data.data();
        eng.plot(last.pixel_selected[0] * last.scaling, last.pixel_selected[1] * last.scaling, po);
    }
    // the regions' maps (math_tools.rs:473-543): what the optical-properties block reads from the last container
    last.roi_signal_fft.clear();
    last.roi_phase_fft.clear();
    last.roi_data.clear();
    for (const auto &kv : last.rois) {
        if (!kv.second.second) continue;
        std::vector<float> a(nf), p(nf), d(nto);
        thz_roi_out ro{};
        ro.signal_fft = a.data(); ro.phase_fft = p.data(); ro.roi_data = d.data();
        if (!eng.roi(kv.first, ro)) continue;
        last.roi_signal_fft[kv.first] = {kv.second.first, a};
        last.roi_phase_fft[kv.first] = {kv.second.first, p};
        last.roi_data[kv.first] = {kv.second.first, d};
    }
    return true;
}

}  // namespace math_tools_gpu

// ------------------------------------------------------------------ plugins on the engine
static GpuEngine *g_walk_engine = nullptr;  // the engine of the pipeline whose walk is running (process-global in Rust)
static GpuEngine &walk_engine() { return g_walk_engine ? *g_walk_engine : GpuEngine::instance(); }

ScannedImageFilterData GpuTiltCompensation::filter(const ScannedImageFilterData &input, GuiSettingsContainer &, ProgressLock &,
                                                   const std::atomic<bool> &)
{
    if (!(input.dx && input.dy)) return input;  // tilt_compensation.rs:111
    GpuEngine &eng = walk_engine();
    eng.record_tilt(true, tilt_x, tilt_y);
    ScannedImageFilterData output = math_tools_gpu::shallow_clone(input);
    // the extended axis (tilt_compensation.rs:104-170, 206-217): the containers behind this stage carry it
    const size_t nt = input.nt();
    const size_t steps = thz_host_tilt_plan(input.time.data(), nt, input.width, input.height, tilt_x, tilt_y, *input.dx, *input.dy,
                                            nullptr, nullptr);
    if (steps) {
        std::vector<float> new_time(nt + 2 * steps);
        thz_host_tilt_plan(input.time.data(), nt, input.width, input.height, tilt_x, tilt_y, *input.dx, *input.dy, new_time.data(), nullptr);
        output.time = new_time;
        output.frequency.resize(new_time.size() / 2 + 1);
        thz_host_frequency_axis(new_time.data(), new_time.size(), output.frequency.data());
    }
    return output;
}

// selected pixel's trace of the current results, for the Time Band Pass plots (after the flush: deferred show_data)
static std::vector<float> plot_filtered_signal(const ScannedImageFilterData &data)
{
    std::vector<float> v(walk_engine().nt_out(), 0.0f);
    thz_plot_out po{};
    po.filtered_signal = v.data();
    walk_engine().plot(data.pixel_selected[0] * data.scaling, data.pixel_selected[1] * data.scaling, po);
    return v;
}
void GpuTimeDomainBandPassBeforeFFT::show_data(const ScannedImageFilterData &data)
{
    if (data.width == 0 || data.height == 0 || data.nt() == 0) return;
    time_axis = data.time;
    // the reference plots the stage's own output — the selected pixel's trace behind the Tilt taper and this band pass.
    // The engine keeps no per-stage cubes; for one trace the stage is two host multiplies on the raw trace (a tilted
    // or scaled chain shows the raw trace itself: its stage output sits on another axis / grid)
    GpuEngine &eng = walk_engine();
    std::vector<float> v(eng.nt, 0.0f);
    thz_plot_out po{};
    po.signal = v.data();
    if (!eng.plot(data.pixel_selected[0] * data.scaling, data.pixel_selected[1] * data.scaling, po)) return;
    if (v.size() == data.nt() && data.scaling <= 1) {
        std::vector<float> w(v.size());
        if (eng.pending().tilt_active) {
            thz_host_adapted_blackman(data.time.data(), v.size(), 0.0f, 7.0f, w.data());
            for (size_t i = 0; i < v.size(); ++i) v[i] *= w[i];
        }
        double lo = low, hi = high;
        thz_host_td_bandpass(data.time.data(), v.size(), &lo, &hi, window_width, w.data(), nullptr, nullptr);
        for (size_t i = 0; i < v.size(); ++i) v[i] *= w[i];
    }
    signal_axis = v;
    input_signal_axis = signal_axis;
}
ScannedImageFilterData GpuTimeDomainBandPassBeforeFFT::filter(const ScannedImageFilterData &input, GuiSettingsContainer &,
                                                              ProgressLock &, const std::atomic<bool> &)
{
    // clamps self.low / self.high like band_pass_td_before_fft.rs:137-138
    std::vector<float> w(input.nt());
    thz_host_td_bandpass(input.time.data(), input.nt(), &low, &high, window_width, w.data(), nullptr, nullptr);
    walk_engine().record_td_before(true, low, high, window_width);
    return math_tools_gpu::shallow_clone(input);
}
void GpuTimeDomainBandPassAfterFFT::show_data(const ScannedImageFilterData &data)
{
    if (data.width == 0 || data.height == 0 || data.nt() == 0) return;
    time_axis = data.time;
    signal_axis = plot_filtered_signal(data);
    input_signal_axis = signal_axis;
}
ScannedImageFilterData GpuTimeDomainBandPassAfterFFT::filter(const ScannedImageFilterData &input, GuiSettingsContainer &,
                                                             ProgressLock &, const std::atomic<bool> &)
{
    std::vector<float> w(input.nt());
    thz_host_td_bandpass(input.time.data(), input.nt(), &low, &high, window_width, w.data(), nullptr, nullptr);
    walk_engine().record_td_after(true, low, high, window_width);
    return math_tools_gpu::shallow_clone(input);
}
void GpuFrequencyDomainBandPass::show_data(const ScannedImageFilterData &data)
{
    if (data.width == 0 || data.height == 0 || data.nf() == 0) return;
    freq_axis = data.frequency;
    std::vector<float> amp(walk_engine().nt_out() / 2 + 1, 0.0f);
    thz_plot_out po{};
    po.filtered_signal_fft = amp.data();  // |band-passed spectrum| of the selected pixel (band_pass_fd.rs show_data)
    if (walk_engine().plot(data.pixel_selected[0] * data.scaling, data.pixel_selected[1] * data.scaling, po)) signal_axis = amp;
}
ScannedImageFilterData GpuFrequencyDomainBandPass::filter(const ScannedImageFilterData &input, GuiSettingsContainer &, ProgressLock &,
                                                          const std::atomic<bool> &)
{
    walk_engine().record_fd(true, low, high, window_width);
    return math_tools_gpu::shallow_clone(input);
}
ScannedImageFilterData GpuDeconvolution::filter(const ScannedImageFilterData &input, GuiSettingsContainer &gui, ProgressLock &progress,
                                                const std::atomic<bool> &abort_flag)
{
    if (!input.dx || !input.dy) {  // deconvolution.rs:781
        log_error("No data loaded, skipping deconvolution.");
        return input;
    }
    const thz_psf psf = gui.psf.view();
    const thz_deconv_cfg cfg{(uint32_t)n_iterations, (uint32_t)n_filters, start_freq, end_freq, win_width, 0u, 0u};
    const int rc = walk_engine().deconvolve(psf, cfg, progress, abort_flag);
    if (rc == THZ_SKIPPED) log_warn("Deconvolution: a guard of the reference applied, input returned unchanged");
    else if (rc < 0) log_error("Deconvolution failed or was aborted (" + walk_engine().last_error() + "), the stage passes its input through");
    return math_tools_gpu::shallow_clone(input);
}
FilterConfig WaterLineNotch::config() const
{
    return {"Water Line Notch", "Suppresses the water vapour absorption lines.", FilterDomain::Frequency};
}
ScannedImageFilterData WaterLineNotch::filter(const ScannedImageFilterData &input, GuiSettingsContainer &, ProgressLock &,
                                              const std::atomic<bool> &)
{
    std::vector<float> m(input.nf(), 1.0f);
    thz_host_water_line_mask(input.frequency.data(), input.nf(), lines_thz.data(), lines_thz.size(), sigma_thz, m.data());
    walk_engine().record_water_lines(true, std::move(m));
    return math_tools_gpu::shallow_clone(input);
}
FilterConfig WienerDeconvolution::config() const
{
    return {"Reference Wiener Filter", "Divides every spectrum by the reference pulse's (Wiener-regularised).",
            FilterDomain::Frequency};
}
ScannedImageFilterData WienerDeconvolution::filter(const ScannedImageFilterData &input, GuiSettingsContainer &, ProgressLock &,
                                                   const std::atomic<bool> &)
{
    const size_t nf = input.nf();
    if (reference_spectrum.size() != 2 * nf) {
        log_warn("Wiener deconvolution: no reference pulse of this length, the stage passes its input through");
        walk_engine().record_wiener(false, {});
        return math_tools_gpu::shallow_clone(input);
    }
    std::vector<float> h(2 * nf);
    thz_host_wiener_filter(reference_spectrum.data(), nf, eps_rel, h.data());
    walk_engine().record_wiener(true, std::move(h));
    return math_tools_gpu::shallow_clone(input);
}

// ------------------------------------------------------------------ the patched stage walk
GpuPipeline::GpuPipeline(GpuEngine &e) : eng(e)
{
    auto add = [&](std::unique_ptr<Filter> f) {
        const std::string uuid = "filter-" + std::to_string(filters.size()) + "-" + f->config().name;
        filters.emplace_back(uuid, std::move(f));
    };
    add(std::make_unique<GpuTiltCompensation>());
    add(std::make_unique<GpuTimeDomainBandPassBeforeFFT>());
    add(std::make_unique<GpuFrequencyDomainBandPass>());
    add(std::make_unique<WaterLineNotch>());
    add(std::make_unique<WienerDeconvolution>());
    add(std::make_unique<GpuTimeDomainBandPassAfterFFT>());
    add(std::make_unique<GpuDeconvolution>());
    filter_chain = {"initial"};
    std::vector<std::string> ordered = {"scaling"};
    auto collect = [&](FilterDomain d) {
        for (auto &f : filters)
            if (f.second->config().domain == d) ordered.push_back(f.first);
    };
    collect(FilterDomain::TimeBeforeFFTPrioFirst);
    collect(FilterDomain::TimeBeforeFFT);
    fft_index = ordered.size();
    ordered.push_back("fft");
    collect(FilterDomain::Frequency);
    ordered.push_back("ifft");
    collect(FilterDomain::TimeAfterFFT);
    collect(FilterDomain::TimeAfterFFTPrioLast);
    filter_uuid_to_index["initial"] = 0;
    for (size_t i = 0; i < ordered.size(); ++i) {
        filter_chain.push_back(ordered[i]);
        filter_uuid_to_index[ordered[i]] = i + 1;
    }
    for (auto &f : filters) {  // main.rs:250-261: Deconvolution starts inactive; the two build-defined plugins too
        const std::string n = f.second->config().name;
        filters_active[f.first] = n.find("Deconvolution") == std::string::npos && n != "Water Line Notch" && n != "Reference Wiener Filter";
    }
    filter_data.resize(filter_chain.size());
}

void GpuPipeline::open(const float *cube, size_t nx, size_t ny, const std::vector<float> &time, float dx, float dy)
{
    eng.open_scan(cube, nx, ny, time, dx, dy, true);
    ScannedImageFilterData s;
    s.width = nx; s.height = ny; s.dx = dx; s.dy = dy; s.time = time; s.has_plan = true;
    s.frequency.resize(time.size() / 2 + 1);
    thz_host_frequency_axis(time.data(), time.size(), s.frequency.data());
    for (auto &d : filter_data) d = s;
    reset_filters = true;
}

size_t GpuPipeline::index_of(const std::string &what) const
{
    for (size_t i = 0; i < filter_chain.size(); ++i) {
        if (filter_chain[i] == what) return i;
        for (auto &f : filters)
            if (f.first == filter_chain[i] && (f.second->config().description == what || f.second->config().name == what)) return i;
    }
    return 0;
}
Filter *GpuPipeline::filter_by(const std::string &what)
{
    for (auto &f : filters)
        if (f.second->config().description == what || f.second->config().name == what) return f.second.get();
    return nullptr;
}

// data_thread.rs:1023-1334 as rust/data_thread.patch leaves it
void GpuPipeline::update_filter(size_t start_idx)
{
    if (start_idx < 1) start_idx = 1;
    g_walk_engine = &eng;
    if (reset_filters) {  // :1027-1060
        for (size_t i = 0; i < filter_chain.size(); ++i) {
            const size_t input_index = i == 0 ? 0 : filter_uuid_to_index[filter_chain[i - 1]];
            for (auto &f : filters)
                if (f.first == filter_chain[i]) {
                    const auto &d = filter_data[input_index];
                    f.second->reset(d.time, {d.width, d.height, d.nt()});
                }
        }
    }
    reset_filters = false;
    std::vector<std::pair<std::string, std::unique_ptr<Filter>>> cloned;  // :1064-1078
    for (auto &f : filters) cloned.emplace_back(f.first, f.second->clone_box());
    bool run_deconvolution = true;
    ProgressLock progress = std::make_shared<std::pair<std::mutex, std::optional<float>>>();
    // PATCH: the engine learns where the walk starts; show_data waits until the results exist
    {
        const std::string &first = filter_chain[start_idx < filter_chain.size() ? start_idx : filter_chain.size() - 1];
        int pos = chain_position(first);
        if (!pos)
            for (auto &f : cloned)
                if (f.first == first) pos = chain_position(f.second->config().domain);
        eng.begin_walk(pos ? pos : 1);
    }
    std::vector<std::pair<Filter *, size_t>> deferred_show_data;
    for (size_t i = start_idx; i < filter_chain.size(); ++i) {
        const std::string &id = filter_chain[i];
        const size_t out_idx = filter_uuid_to_index[id];
        const size_t in_idx = filter_uuid_to_index[filter_chain[i - 1]];
        if (filter_data[in_idx].time.empty()) continue;  // :1099-1105
        if (id == "scaling") {
            filter_data[out_idx] = math_tools_gpu::scaling(eng, filter_data[in_idx], config);
        } else if (id == "fft") {
            filter_data[out_idx] = math_tools_gpu::fft(eng, filter_data[in_idx], config);
        } else if (id == "ifft") {
            filter_data[out_idx] = math_tools_gpu::ifft(eng, filter_data[in_idx], config);
        } else {
            for (auto &f : cloned) {
                if (f.first != id) continue;
                const bool active = filters_active.count(id) ? filters_active[id] : false;
                const bool deconvolution = f.second->config().name.find("Deconvolution") != std::string::npos;
                if (!deconvolution) run_deconvolution = false;  // :1144-1147
                if (active && !(deconvolution && !run_deconvolution)) {
                    filter_data[out_idx] = f.second->filter(filter_data[in_idx], gui_settings, progress, abort_flag);
                    deferred_show_data.emplace_back(f.second.get(), out_idx);  // PATCH (was: show_data right here, :1164)
                } else {
                    filter_data[out_idx] = filter_data[in_idx];
                    eng.note_inactive(f.second->config());                     // PATCH
                }
            }
        }
        // :1194-1227: a stage that changed the axis length re-plans; the big arrays are not re-created (PATCH)
        if (filter_data[in_idx].nt() != filter_data[out_idx].nt()) {
            auto &o = filter_data[out_idx];
            o.frequency.resize(o.nt() / 2 + 1);
            thz_host_frequency_axis(o.time.data(), o.nt(), o.frequency.data());
            o.has_plan = true;
        }
    }
    // PATCH: where the reference sums the image (:1242-1308): one recompute, then what the code behind reads
    math_tools_gpu::finish_stage_walk(eng, filter_data.back(), config);
    for (auto &d : deferred_show_data) d.first->show_data(filter_data[d.second]);
    // :1322-1334 copy the static fields back into the registry
    for (auto &c : cloned)
        for (auto &f : filters)
            if (f.first == c.first) f.second->copy_static_fields_from(*c.second);
    g_walk_engine = nullptr;
}

}  // namespace thzhost
