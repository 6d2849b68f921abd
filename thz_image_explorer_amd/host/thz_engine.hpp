// thz_engine.hpp — the record-and-flush engine the reference-side binding is built around, in C++.
//
// rust/engine.rs, rust/math_tools_gpu.rs and rust/filters/*.rs are the same code in the reference's
// language; no Rust toolchain exists in the build image, so the logic a maintainer would otherwise have to
// take on trust is written — and run, tests/test_gpu_engine.py — here first, and the Rust files are its
// transliteration (same names, same order of calls).
//
// The reference walks its stage chain one stage at a time, each stage returning a complete new container
// (data_thread.rs:1090-1191).  The device engine computes the whole chain in one launch, so on this path a stage
// call only RECORDS its parameters (or, for a plugin the walk passes through, its inactivity) and returns a
// container without the big arrays; after the loop one `flush()` runs `thz_group_session_recompute` from the
// lowest chain position the walk touched, and everything the code behind the loop reads — image, averages,
// per-region means, the selected pixel's traces, each filter's `show_data` plot — is fetched from the device then.
#pragma once

#include "thz_host.hpp"

namespace thzhost {

// chain positions of thz_session_recompute_from: the reference's filter_chain (main.rs:182-247) with all
// FilterDomain::Frequency plugins on one position
enum ChainPosition { kPosScaling = 1, kPosTilt = 2, kPosTdBefore = 3, kPosFft = 4, kPosFrequency = 5, kPosIfft = 6, kPosTdAfter = 7, kPosDeconvolution = 8 };
int chain_position(FilterDomain d);
int chain_position(const std::string &hard_wired_id);  // "scaling" / "fft" / "ifft", else 0

class GpuEngine {
public:
    // devices: the GPUs to tile the x rows over (THZGPU_DEVICES=0,1,... for the process-global instance; the same
    // device several times = slabs side by side on one GPU, what the tests use)
    explicit GpuEngine(const std::vector<int> &devices);
    ~GpuEngine();
    GpuEngine(const GpuEngine &) = delete;
    GpuEngine &operator=(const GpuEngine &) = delete;
    static GpuEngine &instance();
    bool available() const { return group_ != nullptr; }
    std::string last_error() const;

    // ConfigCommand::OpenFile (io.rs:576-628): the cube goes to the device(s) once
    bool open_scan(const float *cube, size_t nx, size_t ny, const std::vector<float> &time, float dx, float dy,
                   bool subtract_bias = true);

    // ---- the stage walk of UpdateType::Filter(start_idx)
    void begin_walk(int start_position);
    void record_scaling(size_t scale_factor);
    void record_tilt(bool active, double tilt_x, double tilt_y);
    void record_td_before(bool active, double low, double high, double width);
    void record_fft(int window_type, float lower, float upper);
    void record_fd(bool active, double low, double high, double width);
    void record_water_lines(bool active, std::vector<float> mask);   // K14, real per-bin multiplier (nf)
    void record_wiener(bool active, std::vector<float> cmask);       // K13, complex per-bin multiplier (2 nf)
    void record_ifft(bool avg_in_fourier_space, const std::vector<std::pair<std::string, Polygon>> &rois);
    void record_td_after(bool active, double low, double high, double width);
    // the walk passed an inactive plugin's input through (data_thread.rs:1185-1188): its stage is off in the chain
    void note_inactive(const FilterConfig &cfg);
    // after the loop (and in front of the Deconvolution stage): one recompute from the lowest touched position
    bool flush();
    // the Deconvolution stage over the whole group; progress and abort forwarded live.  THZ_OK / THZ_SKIPPED / < 0
    int deconvolve(const thz_psf &psf, const thz_deconv_cfg &cfg, ProgressLock &progress, const std::atomic<bool> &abort_flag);

    // ---- results of the last flush
    bool image(std::vector<float> &img, size_t &gx, size_t &gy);   // on the outputs' grid (nx / s, ny / s)
    bool averages(std::vector<std::complex<float>> &avg_fft, std::vector<float> &avg_amp, std::vector<float> &avg_phase);
    bool plot(size_t px, size_t py, const thz_plot_out &out);      // px, py: pixel of the RAW grid
    bool roi(const std::string &uuid, const thz_roi_out &out);
    size_t nt_out() const;
    std::vector<float> time_out() const;
    bool download_final(std::vector<float> &cube);                 // the whole final trace cube, rank order (tests)

    const thz_chain_cfg &pending() const { return pending_; }
    int dirty_from() const { return dirty_from_; }
    size_t nx = 0, ny = 0, nt = 0;

private:
    void touch(int position) { if (position < dirty_from_) dirty_from_ = position < 1 ? 1 : position; }
    thz_session *owner_of(size_t px, size_t *local_px) const;
    thz_group *group_ = nullptr;
    thz_group_session *session_ = nullptr;
    thz_chain_cfg pending_{};
    int dirty_from_ = 1;
    std::vector<float> fd_real_, fd_cmask_;  // empty = that plugin is off
    bool plugins_dirty_ = true;
    std::vector<std::pair<std::string, Polygon>> rois_;
    bool rois_dirty_ = false;
};

// ---- math_tools::{scaling, fft, ifft} on the engine (rust/math_tools_gpu.rs) -----------------------------------
namespace math_tools_gpu {
// metadata, axes, regions and plans of `input`; the five big arrays stay empty (they are resident on the device)
ScannedImageFilterData shallow_clone(const ScannedImageFilterData &input);
ScannedImageFilterData scaling(GpuEngine &eng, const ScannedImageFilterData &input, const ConfigContainer &config);
ScannedImageFilterData fft(GpuEngine &eng, const ScannedImageFilterData &input, const ConfigContainer &config);
ScannedImageFilterData ifft(GpuEngine &eng, const ScannedImageFilterData &input, const ConfigContainer &config);
// after the stage loop (data_thread.rs:1229): flush, then fill what the code behind the loop reads from the LAST
// container: img (expanded s x s when scaled, :1243-1285), the three averages, avg_data, and the regions' maps
bool finish_stage_walk(GpuEngine &eng, ScannedImageFilterData &last, const ConfigContainer &config);
}  // namespace math_tools_gpu

// ---- the plugins on the engine (rust/filters/*.rs): same structs, `filter()` records ------------------------
struct GpuTiltCompensation : TiltCompensation {
    ScannedImageFilterData filter(const ScannedImageFilterData &, GuiSettingsContainer &, ProgressLock &,
                                  const std::atomic<bool> &) override;
    std::unique_ptr<Filter> clone_box() const override { return std::make_unique<GpuTiltCompensation>(*this); }
};
struct GpuTimeDomainBandPassBeforeFFT : TimeDomainBandPassBeforeFFT {
    void show_data(const ScannedImageFilterData &) override;
    ScannedImageFilterData filter(const ScannedImageFilterData &, GuiSettingsContainer &, ProgressLock &,
                                  const std::atomic<bool> &) override;
    std::unique_ptr<Filter> clone_box() const override { return std::make_unique<GpuTimeDomainBandPassBeforeFFT>(*this); }
};
struct GpuTimeDomainBandPassAfterFFT : TimeDomainBandPassAfterFFT {
    void show_data(const ScannedImageFilterData &) override;
    ScannedImageFilterData filter(const ScannedImageFilterData &, GuiSettingsContainer &, ProgressLock &,
                                  const std::atomic<bool> &) override;
    std::unique_ptr<Filter> clone_box() const override { return std::make_unique<GpuTimeDomainBandPassAfterFFT>(*this); }
};
struct GpuFrequencyDomainBandPass : FrequencyDomainBandPass {
    void show_data(const ScannedImageFilterData &) override;
    ScannedImageFilterData filter(const ScannedImageFilterData &, GuiSettingsContainer &, ProgressLock &,
                                  const std::atomic<bool> &) override;
    std::unique_ptr<Filter> clone_box() const override { return std::make_unique<GpuFrequencyDomainBandPass>(*this); }
};
struct GpuDeconvolution : Deconvolution {
    ScannedImageFilterData filter(const ScannedImageFilterData &, GuiSettingsContainer &, ProgressLock &,
                                  const std::atomic<bool> &) override;
    std::unique_ptr<Filter> clone_box() const override { return std::make_unique<GpuDeconvolution>(*this); }
};
// K14 / K13 (build-defined)
struct WaterLineNotch : Filter {
    float sigma_thz = 0.01f;
    std::vector<float> lines_thz;
    FilterConfig config() const override;
    ScannedImageFilterData filter(const ScannedImageFilterData &, GuiSettingsContainer &, ProgressLock &,
                                  const std::atomic<bool> &) override;
    std::unique_ptr<Filter> clone_box() const override { return std::make_unique<WaterLineNotch>(*this); }
};
struct WienerDeconvolution : Filter {
    float eps_rel = 1e-2f;
    std::vector<float> reference_spectrum;  // (nf, 2) interleaved: what OpenRef leaves (thz_reference_spectrum)
    FilterConfig config() const override;
    ScannedImageFilterData filter(const ScannedImageFilterData &, GuiSettingsContainer &, ProgressLock &,
                                  const std::atomic<bool> &) override;
    std::unique_ptr<Filter> clone_box() const override { return std::make_unique<WienerDeconvolution>(*this); }
};

// ---- the patched stage walk (rust/data_thread.patch) ---------------------------------------------------------
// data_thread.rs:1023-1334 with the engine in it: begin_walk at the top, note_inactive where the reference clones an
// inactive plugin's input, show_data deferred until the results exist, finish_stage_walk where the image was summed.
struct GpuPipeline {
    explicit GpuPipeline(GpuEngine &eng);
    GpuEngine &eng;
    std::vector<std::pair<std::string, std::unique_ptr<Filter>>> filters;  // this pipeline's registry: (uuid, filter)
    std::vector<std::string> filter_chain;
    std::map<std::string, size_t> filter_uuid_to_index;
    std::map<std::string, bool> filters_active;
    std::vector<ScannedImageFilterData> filter_data;
    ConfigContainer config;
    GuiSettingsContainer gui_settings;
    std::atomic<bool> abort_flag{false};
    bool reset_filters = true;
    size_t fft_index = 0;

    void open(const float *cube, size_t nx, size_t ny, const std::vector<float> &time, float dx, float dy);
    void update_filter(size_t start_idx);                 // UpdateType::Filter(start_idx)
    size_t index_of(const std::string &description_or_name) const;   // chain index of a filter
    Filter *filter_by(const std::string &description_or_name);
};

}  // namespace thzhost
