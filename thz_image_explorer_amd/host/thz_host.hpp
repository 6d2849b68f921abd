// thz_host.hpp — C++ host-side mirror of the reference's operator / plugin
// surface for the recompute path, on top of the C ABI (include/thzgpu.h).
//
// The reference is Rust and no Rust toolchain exists in the build image, so
// the host side above the C ABI is written in C++ with the reference's names,
// argument meaning and error behaviour:
//
//   ScannedImageFilterData   src/data_container.rs:109-195
//   ConfigContainer          src/config.rs:171-213
//   math_tools::{scaling, fft, ifft}          src/math_tools.rs:242, 330, 418
//   Filter / FilterDomain / FilterConfig / FilterRegistry / register_filter
//                            src/filters/filter.rs:96-455, filter_macros/src/lib.rs:5-43
//   the five filters         src/filters/{tilt_compensation,band_pass_td_before_fft,
//                            band_pass_fd,band_pass_td_after_fft,deconvolution}.rs
//   Pipeline                 stage order src/main.rs:194-268, stage walk
//                            src/data_thread.rs:1023-1334
//
// Differences that follow from the device: the big arrays of a container live
// in HBM (DeviceArray; `clone()` is a device-to-device copy, exactly where the
// reference deep-clones an ndarray); host-visible vectors (time, frequency,
// averages, ROI traces) stay std::vectors.  Errors follow the reference: log
// and return the input unchanged.
#pragma once

#include "../../include/thzgpu.h"

#include <array>
#include <atomic>
#include <complex>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <string>
#include <utility>
#include <vector>

namespace thzhost {

// ---- engine handle: process-global, like the once_cell::Lazy of INTEGRATION.md §3
class Engine {
public:
    static Engine &instance();
    thz_ctx *ctx() { return ctx_; }
    // (re)plans when the trace length / axis differs from the current one
    bool ensure_axis(const std::vector<float> &time);
    std::string last_error() const;
    std::mutex mu;  // one stage call at a time (the reference's single data thread)
private:
    Engine();
    ~Engine();
    thz_ctx *ctx_ = nullptr;
    std::vector<float> axis_;
};

// ---- device-resident array with value semantics (copy = device-to-device)
class DeviceArray {
public:
    DeviceArray() = default;
    explicit DeviceArray(size_t n_floats) { resize(n_floats); }
    DeviceArray(const DeviceArray &o);
    DeviceArray &operator=(const DeviceArray &o);
    DeviceArray(DeviceArray &&o) noexcept { swap(o); }
    DeviceArray &operator=(DeviceArray &&o) noexcept { swap(o); return *this; }
    ~DeviceArray();
    void resize(size_t n_floats);  // contents undefined
    void zero();
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    float *ptr() { return p_; }
    const float *ptr() const { return p_; }
    void upload(const float *src, size_t n);
    std::vector<float> download() const;
    std::vector<float> download(size_t offset, size_t n) const;
    void swap(DeviceArray &o) noexcept { std::swap(p_, o.p_); std::swap(n_, o.n_); }
private:
    float *p_ = nullptr;
    size_t n_ = 0;
};

using Polygon = std::vector<std::pair<size_t, size_t>>;

// data_container.rs:109-162
struct ScannedImageFilterData {
    std::optional<float> x_min, dx, y_min, dy;
    size_t height = 0, width = 0;
    size_t scaling = 1;
    std::array<size_t, 2> pixel_selected{0, 0};
    bool has_plan = false;  // r2c / c2r: Some(..) once a file is loaded (io.rs:616-624)
    std::map<std::string, std::pair<std::string, std::optional<Polygon>>> rois;
    std::vector<float> time;
    DeviceArray img;         // (width, height)
    DeviceArray data;        // (width, height, nt)
    std::vector<float> avg_data;
    std::map<std::string, std::pair<std::string, std::vector<float>>> roi_data;
    std::vector<float> frequency;
    DeviceArray fft;         // (width, height, nf) complex interleaved
    DeviceArray amplitudes;  // (width, height, nf)
    DeviceArray phases;      // (width, height, nf)
    std::vector<std::complex<float>> avg_fft;
    std::vector<float> avg_signal_fft, avg_phase_fft;
    std::map<std::string, std::pair<std::string, std::vector<float>>> roi_signal_fft, roi_phase_fft;

    size_t nt() const { return time.size(); }
    size_t nf() const { return frequency.size(); }
    size_t npix() const { return width * height; }
    // io.rs:496-631 equivalent for an in-memory cube: bias subtraction, image,
    // frequency axis, zeroed spectra
    static ScannedImageFilterData from_host_cube(const float *cube, size_t width, size_t height,
                                                 const std::vector<float> &time, float dx, float dy);
};

// io.rs:496-631 through libthzio.so (include/thzio.h; loaded at run time, so the mirror
// itself has no HDF5 dependency): the cube goes to the device in x-slabs of `slab_rows`
// rows (0 = about 256 MiB each) — a reader thread fills the next slab while the current
// one is copied — then bias subtraction + image run once on the device.  Throws
// std::runtime_error when the file cannot be read.  metadata: mdDescription -> md values.
namespace io {
ScannedImageFilterData open_scan_from_thz(const std::string &path, std::map<std::string, std::string> *metadata = nullptr,
                                          size_t slab_rows = 0);
// open_pulse_from_thz, io.rs:435-477: (time, signal); both empty when the first dataset is not 2-D
std::pair<std::vector<float>, std::vector<float>> open_pulse_from_thz(const std::string &path);
}  // namespace io

// math_tools.rs:35-46
enum class FftWindowType { AdaptedBlackman = 0, Blackman = 1, Hanning = 2, Hamming = 3, FlatTop = 4 };

// config.rs:171-213
struct ConfigContainer {
    std::array<float, 2> fft_window{1.0f, 7.0f};
    FftWindowType fft_window_type = FftWindowType::AdaptedBlackman;
    size_t scale_factor = 1;
    bool fft_log_plot = false;
    bool avg_in_fourier_space = false;
    float fft_df = 1.0f;
};

namespace math_tools {
ScannedImageFilterData scaling(const ScannedImageFilterData &input, const ConfigContainer &config);
ScannedImageFilterData fft(const ScannedImageFilterData &input, const ConfigContainer &config);
ScannedImageFilterData ifft(const ScannedImageFilterData &input, const ConfigContainer &config);
}  // namespace math_tools

// ---- plugin surface, filters/filter.rs:96-455 ---------------------------------
enum class FilterDomain { TimeBeforeFFTPrioFirst, TimeBeforeFFT, Frequency, TimeAfterFFT, TimeAfterFFTPrioLast };

struct FilterConfig {
    std::string name, description;
    FilterDomain domain;
};

// the parts of GuiSettingsContainer a filter reads (gui/application.rs:134-218)
struct PsfArrays {  // owns the arrays a thz_psf points into
    float wx_a = 0, wx_b = 0, wy_a = 0, wy_b = 0;
    std::vector<float> k[4], v[4], a[4], b[4], c[4], d[4];  // wx, wy, x0, y0
    thz_psf view() const;
};
struct GuiSettingsContainer {
    PsfArrays psf;
};
using ProgressLock = std::shared_ptr<std::pair<std::mutex, std::optional<float>>>;

class Filter {
public:
    virtual ~Filter() = default;
    virtual void reset(const std::vector<float> &time, const std::array<size_t, 3> &shape) { (void)time; (void)shape; }
    virtual void show_data(const ScannedImageFilterData &) {}
    virtual FilterConfig config() const = 0;
    virtual ScannedImageFilterData filter(const ScannedImageFilterData &input, GuiSettingsContainer &gui_settings,
                                          ProgressLock &progress_lock, const std::atomic<bool> &abort_flag) = 0;
    virtual std::unique_ptr<Filter> clone_box() const = 0;                 // CloneBoxedFilter
    virtual void copy_static_fields_from(const Filter &) {}               // CopyStaticFieldsTrait
};

class FilterRegistry {
public:
    static FilterRegistry &global();
    template <class F>
    static void register_filter()  // what #[register_filter] expands to (filter_macros/src/lib.rs:33-40)
    {
        global().add(std::make_unique<F>());
    }
    void add(std::unique_ptr<Filter> f);
    std::vector<std::pair<std::string, std::unique_ptr<Filter>>> filters;  // (uuid, filter)
    std::mutex mu;
};
#define THZ_REGISTER_FILTER(T) \
    static const bool thz_registered_##T = (::thzhost::FilterRegistry::register_filter<T>(), true)

// ---- the reference's filters ----------------------------------------------------
struct TiltCompensation : Filter {  // tilt_compensation.rs
    double tilt_x = 0.0, tilt_y = 0.0;
    FilterConfig config() const override;
    ScannedImageFilterData filter(const ScannedImageFilterData &, GuiSettingsContainer &, ProgressLock &,
                                  const std::atomic<bool> &) override;
    std::unique_ptr<Filter> clone_box() const override { return std::make_unique<TiltCompensation>(*this); }
};

struct TimeDomainBandPassBeforeFFT : Filter {  // band_pass_td_before_fft.rs
    double low = 0.0, high = 0.0, window_width = 2.0;
    std::vector<float> time_axis, signal_axis, input_signal_axis;  // #[static_field]
    void reset(const std::vector<float> &time, const std::array<size_t, 3> &shape) override;
    void show_data(const ScannedImageFilterData &) override;
    FilterConfig config() const override;
    ScannedImageFilterData filter(const ScannedImageFilterData &, GuiSettingsContainer &, ProgressLock &,
                                  const std::atomic<bool> &) override;
    std::unique_ptr<Filter> clone_box() const override { return std::make_unique<TimeDomainBandPassBeforeFFT>(*this); }
    void copy_static_fields_from(const Filter &o) override;
};

struct TimeDomainBandPassAfterFFT : TimeDomainBandPassBeforeFFT {  // band_pass_td_after_fft.rs
    TimeDomainBandPassAfterFFT() { window_width = 0.1; }
    FilterConfig config() const override;
    std::unique_ptr<Filter> clone_box() const override { return std::make_unique<TimeDomainBandPassAfterFFT>(*this); }
};

struct FrequencyDomainBandPass : Filter {  // band_pass_fd.rs
    double low = 0.2, high = 5.0, window_width = 0.1;
    std::vector<float> freq_axis, signal_axis;  // #[static_field]
    void show_data(const ScannedImageFilterData &) override;
    FilterConfig config() const override;
    ScannedImageFilterData filter(const ScannedImageFilterData &, GuiSettingsContainer &, ProgressLock &,
                                  const std::atomic<bool> &) override;
    std::unique_ptr<Filter> clone_box() const override { return std::make_unique<FrequencyDomainBandPass>(*this); }
    void copy_static_fields_from(const Filter &o) override;
};

struct Deconvolution : Filter {  // deconvolution.rs:239-253, 715-1041
    size_t n_iterations = 500, n_filters = 25;
    float start_freq = 0.1f, end_freq = 10.0f, win_width = 0.5f;
    FilterConfig config() const override;
    ScannedImageFilterData filter(const ScannedImageFilterData &, GuiSettingsContainer &, ProgressLock &,
                                  const std::atomic<bool> &) override;
    std::unique_ptr<Filter> clone_box() const override { return std::make_unique<Deconvolution>(*this); }
};

// ---- stage order + stage walk -----------------------------------------------------
struct Pipeline {
    // main.rs:178-268: "initial", "scaling", [PrioFirst], [TimeBeforeFFT], "fft",
    // [Frequency], "ifft", [TimeAfterFFT], [PrioLast]
    Pipeline();
    std::vector<std::string> filter_chain;
    std::map<std::string, size_t> filter_uuid_to_index;
    std::map<std::string, bool> filters_active;
    std::vector<ScannedImageFilterData> filter_data;
    std::map<std::string, double> filter_computation_time_ms;
    size_t fft_index = 0, ifft_index = 0;
    ConfigContainer config;
    GuiSettingsContainer gui_settings;
    std::atomic<bool> abort_flag{false};
    bool reset_filters = true;

    void open(ScannedImageFilterData scan);  // OpenFile: slot 0 <- scan, reset_filters
    // UpdateType::Filter(start_idx), data_thread.rs:1023-1334
    void update_filter(size_t start_idx);
    std::string uuid_of(const std::string &filter_name) const;
};

}  // namespace thzhost
