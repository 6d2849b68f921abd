//! "Frequency Band Pass" (`src/filters/band_pass_fd.rs`): same struct, UI and config; `filter()` hands its three
//! numbers to the engine, whose fused launch applies the multiplier (index rule `band_pass_fd.rs:135-168`,
//! zero padding `:194-212` — `thz_host_fd_bandpass`).  UNVERIFIED BY A COMPILER.
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
pub struct FrequencyDomainBandPass {
    pub low: f64,
    pub high: f64,
    pub window_width: f64,
    #[static_field]
    freq_axis: Vec<f32>,
    #[static_field]
    signal_axis: Vec<f32>,
}

impl Filter for FrequencyDomainBandPass {
    fn new() -> Self where Self: Sized {
        FrequencyDomainBandPass { low: 0.2, high: 5.0, window_width: 0.1, freq_axis: vec![], signal_axis: vec![] }
    }

    fn reset(&mut self, _time: &Array1<f32>, _shape: &[usize]) {}

    fn show_data(&mut self, data: &ScannedImageFilterData) {
        // the plot of the selected pixel's spectrum: one nf-vector from the device instead of a slice of a host cube
        self.freq_axis = data.frequency.to_vec();
        let mut amp = vec![0f32; data.frequency.len()];
        let out = crate::gpu::ffi::ThzPlotOut {
            signal: std::ptr::null_mut(), signal_fft: std::ptr::null_mut(), phase_fft: std::ptr::null_mut(),
            filtered_signal: std::ptr::null_mut(), filtered_signal_fft: amp.as_mut_ptr(),
            filtered_phase_fft: std::ptr::null_mut(), avg_signal: std::ptr::null_mut(),
            avg_signal_fft: std::ptr::null_mut(), avg_phase_fft: std::ptr::null_mut(),
        };
        if ENGINE.lock().unwrap().plot(data.pixel_selected[0], data.pixel_selected[1], &out).is_ok() {
            self.signal_axis = amp;
        }
    }

    fn config(&self) -> FilterConfig {
        FilterConfig { name: "Frequency Band Pass".to_string(), description: "Band Pass Filter in Frequency Domain.".to_string(),
                       hyperlink: None, domain: FilterDomain::Frequency }
    }

    fn filter(&mut self, input_data: &ScannedImageFilterData, _gui_settings: &mut GuiSettingsContainer,
              _progress_lock: &mut Arc<RwLock<Option<f32>>>, _abort_flag: &Arc<AtomicBool>) -> ScannedImageFilterData {
        let mut eng = ENGINE.lock().unwrap();
        if !eng.available() { return input_data.clone(); }
        eng.record_fd(true, self.low, self.high, self.window_width);
        shallow_clone(input_data)
    }

    fn ui(&mut self, ui: &mut Ui, _thread_communication: &mut ThreadCommunication, _panel_width: f32) -> egui::Response {
        // unchanged from the reference (band_pass_fd.rs:222-…): double slider over freq_axis + spectrum plot
        crate::filters::band_pass_fd_ui::draw(self_low_high_width(self), &self.freq_axis, &self.signal_axis, ui)
    }
}

fn self_low_high_width(f: &mut FrequencyDomainBandPass) -> (&mut f64, &mut f64, &mut f64) {
    (&mut f.low, &mut f.high, &mut f.window_width)
}
