//! K13 — reference-pulse Wiener deconvolution (build-defined: the reference has no such filter; BASELINE config
//! 5 names it).  `FilterDomain::Frequency`, behind "Frequency Band Pass".
//!     H[k] = conj(R[k]) / (|R[k]|^2 + eps * max_j |R[j]|^2)
//! with R the spectrum of the reference pulse as `ConfigCommand::OpenRef` ingests it (aligned to the scan's
//! axis, windowed with the reference file's own axis, transformed with the scan's plan: `data_thread.rs:405-533`
//! = `thz_reference_spectrum`).  The complex per-bin multiplier rides inside the engine's fused launch
//! (imaginary parts of bin 0 / Nyquist forced to 0: realfft's C2R precondition, `math_tools.rs:510-512`).
//! The plugin's NAME must not contain "Deconvolution": the data thread treats every filter whose name does as the
//! Richardson-Lucy stage that only runs when it is the one being updated (`data_thread.rs:1139-1149`).
//! Transliteration of `WienerDeconvolution` (`thz_image_explorer_amd/host/thz_engine.cpp`).  A complete new file: list it
//! in `src/filters/mod.rs:23-47`.  UNVERIFIED BY A COMPILER.
use crate::config::ThreadCommunication;
use crate::data_container::ScannedImageFilterData;
use crate::filters::filter::{CopyStaticFieldsTrait, Filter, FilterConfig, FilterDomain};
use crate::gpu::engine::ENGINE;
use crate::gpu::ffi::thz_host_wiener_filter;
use crate::gui::application::GuiSettingsContainer;
use crate::math_tools_gpu::shallow_clone;
use bevy_egui::egui::{self, Ui};
use filter_macros::{register_filter, CopyStaticFields};
use ndarray::Array1;
use std::sync::atomic::AtomicBool;
use std::sync::{Arc, RwLock};

#[register_filter]
#[derive(Clone, Debug, CopyStaticFields)]
pub struct WienerDeconvolution {
    /// regularisation relative to the strongest bin of the reference
    pub eps_rel: f32,
    /// key of the reference pulse in `ScannedImageFilterData::roi_signal_fft` / `roi_phase_fft` (what OpenRef fills)
    pub reference_key: String,
}

impl Filter for WienerDeconvolution {
    fn new() -> Self where Self: Sized { WienerDeconvolution { eps_rel: 1e-2, reference_key: "Reference".to_string() } }
    fn reset(&mut self, _time: &Array1<f32>, _shape: &[usize]) {}
    fn show_data(&mut self, _data: &ScannedImageFilterData) {}

    fn config(&self) -> FilterConfig {
        FilterConfig { name: "Reference Wiener Filter".to_string(),
                       description: "Divides every spectrum by the reference pulse's (Wiener-regularised).".to_string(),
                       hyperlink: None, domain: FilterDomain::Frequency }
    }

    fn filter(&mut self, input_data: &ScannedImageFilterData, _gui_settings: &mut GuiSettingsContainer,
              _progress_lock: &mut Arc<RwLock<Option<f32>>>, _abort_flag: &Arc<AtomicBool>) -> ScannedImageFilterData {
        let mut eng = ENGINE.lock().unwrap();
        if !eng.available() { return input_data.clone(); }
        let nf = input_data.frequency.len();
        // R[k] = amp[k] exp(i phase[k]) from OpenRef's vectors (length nt, the first nf filled: data_thread.rs:486-533)
        let (amp, ph) = match (input_data.roi_signal_fft.get(&self.reference_key), input_data.roi_phase_fft.get(&self.reference_key)) {
            (Some((_, a)), Some((_, p))) if a.len() >= nf && p.len() >= nf => (a, p),
            _ => {
                log::warn!("Wiener filter: no reference pulse loaded, the stage passes its input through");
                eng.record_wiener(false, vec![]);
                return shallow_clone(input_data);
            }
        };
        let mut r = vec![0f32; 2 * nf];
        for k in 0..nf {
            r[2 * k] = amp[k] * ph[k].cos();
            r[2 * k + 1] = amp[k] * ph[k].sin();
        }
        let mut h = vec![0f32; 2 * nf];
        unsafe { thz_host_wiener_filter(r.as_ptr(), nf, self.eps_rel, h.as_mut_ptr()); }
        eng.record_wiener(true, h);
        shallow_clone(input_data)
    }

    fn ui(&mut self, ui: &mut Ui, _thread_communication: &mut ThreadCommunication, _panel_width: f32) -> egui::Response {
        ui.add(egui::Slider::new(&mut self.eps_rel, 1e-4..=1.0).logarithmic(true).text("regularisation"))
    }
}
