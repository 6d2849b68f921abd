//! "Time Band Pass" (`src/filters/band_pass_td_before_fft.rs`): same struct, UI and config; `filter()` records low / high / width
//! (clamped like `:137-138`) and the engine multiplies by the vector of `thz_host_td_bandpass` (zero outside
//! [lower, upper), adapted-Blackman taper inside, `:124-182`) — inside the fused launch.  UNVERIFIED BY A COMPILER.
use crate::config::ThreadCommunication;
use crate::data_container::ScannedImageFilterData;
use crate::filters::filter::{CopyStaticFieldsTrait, Filter, FilterConfig, FilterDomain};
use crate::gpu::engine::ENGINE;
use crate::gui::application::GuiSettingsContainer;
use crate::math_tools_gpu::shallow_clone;
use bevy_egui::egui::{self, Ui};
use filter_macros::{register_filter, CopyStaticFields};
use ndarray::Array1;
use std::sync::atomic::AtomicBool;
use std::sync::{Arc, RwLock};

#[register_filter]
#[derive(Clone, Debug, CopyStaticFields)]
pub struct TimeDomainBandPassBeforeFFT {
    pub low: f64,
    pub high: f64,
    pub window_width: f64,
    #[static_field]
    time_axis: Vec<f32>,
    #[static_field]
    signal_axis: Vec<f32>,
    #[static_field]
    input_signal_axis: Vec<f32>,
}

impl Filter for TimeDomainBandPassBeforeFFT {
    fn new() -> Self where Self: Sized {
        TimeDomainBandPassBeforeFFT { low: 0.0, high: 0.0, window_width: 2.0, time_axis: vec![], signal_axis: vec![], input_signal_axis: vec![] }
    }

    /// full range of the new file's axis (band_pass_td_before_fft.rs:66-72)
    fn reset(&mut self, time: &Array1<f32>, _shape: &[usize]) {
        if let (Some(first), Some(last)) = (time.first(), time.last()) {
            self.low = *first as f64;
            self.high = *last as f64;
        }
        self.time_axis = time.to_vec();
    }

    fn show_data(&mut self, data: &ScannedImageFilterData) {
        self.time_axis = data.time.to_vec();
        let mut trace = vec![0f32; data.time.len()];
        let out = crate::gpu::ffi::ThzPlotOut {
            signal: std::ptr::null_mut(), signal_fft: std::ptr::null_mut(), phase_fft: std::ptr::null_mut(),
            filtered_signal: trace.as_mut_ptr(), filtered_signal_fft: std::ptr::null_mut(),
            filtered_phase_fft: std::ptr::null_mut(), avg_signal: std::ptr::null_mut(),
            avg_signal_fft: std::ptr::null_mut(), avg_phase_fft: std::ptr::null_mut(),
        };
        if ENGINE.lock().unwrap().plot(data.pixel_selected[0], data.pixel_selected[1], &out).is_ok() {
            self.signal_axis = trace;
        }
    }

    fn config(&self) -> FilterConfig {
        FilterConfig { name: "Time Band Pass".to_string(), description: "Band Pass Filter in Time Domain.".to_string(),
                       hyperlink: None, domain: FilterDomain::TimeBeforeFFT }
    }

    fn filter(&mut self, input_data: &ScannedImageFilterData, _gui_settings: &mut GuiSettingsContainer,
              _progress_lock: &mut Arc<RwLock<Option<f32>>>, _abort_flag: &Arc<AtomicBool>) -> ScannedImageFilterData {
        let mut eng = ENGINE.lock().unwrap();
        if !eng.available() { return input_data.clone(); }
        eng.record_td_before(true, self.low, self.high, self.window_width);
        shallow_clone(input_data)
    }

    fn ui(&mut self, ui: &mut Ui, _thread_communication: &mut ThreadCommunication, _panel_width: f32) -> egui::Response {
        // unchanged from the reference: double slider over time_axis + trace plot
        crate::filters::band_pass_td_ui::draw(&mut self.low, &mut self.high, &mut self.window_width, &self.time_axis,
                                              &self.signal_axis, &self.input_signal_axis, ui)
    }
}
