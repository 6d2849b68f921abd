//! K14 — water-line notch (build-defined: the reference only draws the lines, `gui/center_panel.rs:477-485`;
//! BASELINE config 3 names a water-line filter).  `FilterDomain::Frequency`, behind "Frequency Band Pass":
//!     m[k] = prod_i (1 - exp(-((f_k - line_i) / sigma)^2))   over the 135 lines of assets/water_lines.csv
//! a real per-bin multiplier applied in the engine's fused launch like the band pass itself.
//! Transliteration of `WaterLineNotch` (`thz_image_explorer_amd/host/thz_engine.cpp`, tested by `tests/test_gpu_engine.py`:
//! switched on, the walk's output is the band pass times the notch; switched off — the data thread then calls
//! `note_inactive` instead of `filter()` — it is gone again).  A complete new file: list it in `src/filters/mod.rs:23-47`.
//! UNVERIFIED BY A COMPILER.
use crate::config::ThreadCommunication;
use crate::data_container::ScannedImageFilterData;
use crate::filters::filter::{CopyStaticFieldsTrait, Filter, FilterConfig, FilterDomain};
use crate::gpu::engine::ENGINE;
use crate::gpu::ffi::thz_host_water_line_mask;
use crate::gui::application::GuiSettingsContainer;
use crate::math_tools_gpu::shallow_clone;
use bevy_egui::egui::{self, Ui};
use filter_macros::{register_filter, CopyStaticFields};
use ndarray::Array1;
use std::sync::atomic::AtomicBool;
use std::sync::{Arc, RwLock};

#[register_filter]
#[derive(Clone, Debug, CopyStaticFields)]
pub struct WaterLineNotch {
    /// 1/e half-width of every notch, THz
    pub sigma_thz: f32,
    #[static_field]
    lines_thz: Vec<f32>,
}

impl Filter for WaterLineNotch {
    fn new() -> Self where Self: Sized {
        // the same file the GUI draws its markers from
        let lines = include_str!("../../assets/water_lines.csv").lines().filter_map(|l| l.trim().parse::<f32>().ok()).collect();
        WaterLineNotch { sigma_thz: 0.01, lines_thz: lines }
    }
    fn reset(&mut self, _time: &Array1<f32>, _shape: &[usize]) {}
    fn show_data(&mut self, _data: &ScannedImageFilterData) {}

    fn config(&self) -> FilterConfig {
        FilterConfig { name: "Water Line Notch".to_string(),
                       description: "Suppresses the water vapour absorption lines.".to_string(),
                       hyperlink: None, domain: FilterDomain::Frequency }
    }

    fn filter(&mut self, input_data: &ScannedImageFilterData, _gui_settings: &mut GuiSettingsContainer,
              _progress_lock: &mut Arc<RwLock<Option<f32>>>, _abort_flag: &Arc<AtomicBool>) -> ScannedImageFilterData {
        let mut eng = ENGINE.lock().unwrap();
        if !eng.available() { return input_data.clone(); }
        let f = input_data.frequency.as_slice().unwrap();
        let mut m = vec![1f32; f.len()];
        unsafe { thz_host_water_line_mask(f.as_ptr(), f.len(), self.lines_thz.as_ptr(), self.lines_thz.len(), self.sigma_thz, m.as_mut_ptr()); }
        eng.record_water_lines(true, m);
        shallow_clone(input_data)
    }

    fn ui(&mut self, ui: &mut Ui, _thread_communication: &mut ThreadCommunication, _panel_width: f32) -> egui::Response {
        ui.add(egui::Slider::new(&mut self.sigma_thz, 0.002..=0.1).text("notch width (THz)"))
    }
}
