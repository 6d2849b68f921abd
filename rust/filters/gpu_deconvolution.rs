//! "Deconvolution" (`src/filters/deconvolution.rs:766-1041`): same struct, UI and config; `filter()` runs
//! `thz_session_deconvolve` — FIR bank, per-band energy image, Richardson-Lucy against the band's Gaussian PSF
//! (all bands batched per iteration), gains, ONE recombination transform — on the device.  The guards of the
//! reference (empty PSF, image < 16 x 16, PSF wider than the image: `:790-885`) come back as `THZ_SKIPPED`, an
//! abort as `THZ_ERR_ABORTED`; in both cases the input is what the stage outputs, as in the reference.
//! Progress is written into `progress_lock` while the call runs and the ✕ button's `abort_flag` is seen between
//! iteration batches (`GpuEngine::deconvolve`).  UNVERIFIED BY A COMPILER.
use crate::config::ThreadCommunication;
use crate::data_container::ScannedImageFilterData;
use crate::filters::filter::{CopyStaticFieldsTrait, Filter, FilterConfig, FilterDomain};
use crate::filters::psf::{CubicSplineCoeffs, HybridFit, PSF};
use crate::gpu::engine::ENGINE;
use crate::gpu::ffi::{ThzDeconvCfg, ThzHybridFit, ThzPsf, ThzSpline, THZ_SKIPPED};
use crate::gui::application::GuiSettingsContainer;
use crate::math_tools_gpu::shallow_clone;
use bevy_egui::egui::{self, Ui};
use filter_macros::{register_filter, CopyStaticFields};
use ndarray::Array1;
use std::sync::atomic::AtomicBool;
use std::sync::{Arc, RwLock};

#[register_filter]
#[derive(Clone, Debug, CopyStaticFields)]
pub struct Deconvolution {
    pub n_iterations: usize,
    pub n_filters: usize,
    pub start_freq: f32,
    pub end_freq: f32,
    pub win_width: f32,
    #[static_field]
    expert_mode: bool,
}

fn spline_view(s: &CubicSplineCoeffs) -> ThzSpline {
    ThzSpline { knots: s.knots.as_ptr(), values: s.values.as_ptr(), coeff_a: s.coeff_a.as_ptr(), coeff_b: s.coeff_b.as_ptr(),
                coeff_c: s.coeff_c.as_ptr(), coeff_d: s.coeff_d.as_ptr(), n_knots: s.knots.len() }
}
fn fit_view(f: &HybridFit) -> ThzHybridFit { ThzHybridFit { base_a: f.base_a, base_b: f.base_b, correction: spline_view(&f.correction) } }
/// borrows the PSF's arrays: valid while `psf` lives
pub fn psf_view(psf: &PSF) -> ThzPsf {
    ThzPsf { wx_fit: fit_view(&psf.wx_fit), wy_fit: fit_view(&psf.wy_fit), x0_spline: spline_view(&psf.x0_spline),
             y0_spline: spline_view(&psf.y0_spline) }
}

impl Filter for Deconvolution {
    fn new() -> Self where Self: Sized {
        Deconvolution { n_iterations: 500, n_filters: 25, start_freq: 0.1, end_freq: 10.0, win_width: 0.5, expert_mode: false }
    }
    fn reset(&mut self, _time: &Array1<f32>, _shape: &[usize]) {}
    fn show_data(&mut self, _data: &ScannedImageFilterData) {}

    fn config(&self) -> FilterConfig {
        FilterConfig {
            name: "Deconvolution".to_string(),
            description: "Frequency-dependent deconvolution for enhanced THz-TDS scans, accounting for beam width variations in time traces.".to_string(),
            hyperlink: Some((Some("TTHZ.2025.3546756".to_string()), "https://doi.org/10.1109/TTHZ.2025.3546756".to_string())),
            domain: FilterDomain::TimeAfterFFTPrioLast,
        }
    }

    fn filter(&mut self, input_data: &ScannedImageFilterData, gui_settings: &mut GuiSettingsContainer,
              progress_lock: &mut Arc<RwLock<Option<f32>>>, abort_flag: &Arc<AtomicBool>) -> ScannedImageFilterData {
        let mut eng = ENGINE.lock().unwrap();
        if !eng.available() { return input_data.clone(); }
        // everything in front of this stage must be on the device before it runs
        if let Err(e) = eng.flush() { log::error!("gpu recompute before deconvolution: {}", e.1); return input_data.clone(); }
        let psf = psf_view(&gui_settings.psf);
        let cfg = ThzDeconvCfg { n_iterations: self.n_iterations as u32, n_filters: self.n_filters as u32, start_freq: self.start_freq,
                                 end_freq: self.end_freq, win_width: self.win_width, band_begin: 0, band_end: 0 };
        match eng.deconvolve(&psf, &cfg, progress_lock, abort_flag) {
            Ok(rc) if rc == THZ_SKIPPED => log::warn!("Deconvolution: a guard of the reference applied, input returned unchanged"),
            Ok(_) => {}
            Err(e) => { log::error!("Deconvolution failed ({}: {}), returning the input", e.0, e.1); return input_data.clone(); }
        }
        shallow_clone(input_data)
    }

    fn ui(&mut self, ui: &mut Ui, _thread_communication: &mut ThreadCommunication, _panel_width: f32) -> egui::Response {
        // unchanged from the reference (deconvolution.rs:1043-…): iterations slider, expert mode with the bank parameters
        let mut final_response = ui.allocate_response(egui::Vec2::ZERO, egui::Sense::hover());
        let r = ui.horizontal(|ui| { ui.label("Iterations: "); ui.add(egui::Slider::new(&mut self.n_iterations, 1..=1000)) }).inner;
        final_response |= r.clone();
        ui.checkbox(&mut self.expert_mode, "Expert mode");
        if self.expert_mode {
            let r2 = ui.add(egui::Slider::new(&mut self.n_filters, 2..=50).text("filters"));
            let r3 = ui.add(egui::Slider::new(&mut self.start_freq, 0.05..=1.0).text("start (THz)"));
            let r4 = ui.add(egui::Slider::new(&mut self.end_freq, 2.0..=10.0).text("end (THz)"));
            let r5 = ui.add(egui::Slider::new(&mut self.win_width, 0.1..=2.0).text("window (THz)"));
            if r2.changed() || r3.changed() || r4.changed() || r5.changed() { final_response.mark_changed(); }
        }
        if r.changed() { final_response.mark_changed(); }
        final_response
    }
}
