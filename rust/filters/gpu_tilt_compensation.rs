//! "Tilt Compensation" (`src/filters/tilt_compensation.rs`): same struct, UI and config; `filter()` records the
//! two angles.  Zero tilt is just the tail taper (`:186-188`), which the engine folds into the fused launch's
//! pre-transform multiplier; a non-zero tilt re-lays the cube out on the extended axis on the device
//! (`thz_host_tilt_plan` + `thz_tilt_apply`, geometry `:104-175`, hard-coded dt = 0.05 ps `:122`) and the chain
//! continues at the new length — the container's `time` / `frequency` then come from `thz_session_time_out`.
//! UNVERIFIED BY A COMPILER.
use crate::config::ThreadCommunication;
use crate::data_container::ScannedImageFilterData;
use crate::filters::filter::{CopyStaticFieldsTrait, Filter, FilterConfig, FilterDomain};
use crate::gpu::engine::ENGINE;
use crate::gpu::ffi::thz_host_tilt_plan;
use crate::gui::application::GuiSettingsContainer;
use crate::math_tools_gpu::shallow_clone;
use bevy_egui::egui::{self, Ui};
use filter_macros::{register_filter, CopyStaticFields};
use ndarray::Array1;
use std::sync::atomic::AtomicBool;
use std::sync::{Arc, RwLock};

#[register_filter]
#[derive(Clone, Debug, CopyStaticFields)]
pub struct TiltCompensation {
    pub tilt_x: f64,
    pub tilt_y: f64,
}

impl Filter for TiltCompensation {
    fn new() -> Self where Self: Sized { TiltCompensation { tilt_x: 0.0, tilt_y: 0.0 } }
    fn reset(&mut self, _time: &Array1<f32>, _shape: &[usize]) {}
    fn show_data(&mut self, _data: &ScannedImageFilterData) {}

    fn config(&self) -> FilterConfig {
        FilterConfig { name: "Tilt Compensation".to_string(),
                       description: "Compensate any misalignment along x axis and y axis.".to_string(),
                       hyperlink: None, domain: FilterDomain::TimeBeforeFFTPrioFirst }
    }

    fn filter(&mut self, input_data: &ScannedImageFilterData, _gui_settings: &mut GuiSettingsContainer,
              _progress_lock: &mut Arc<RwLock<Option<f32>>>, _abort_flag: &Arc<AtomicBool>) -> ScannedImageFilterData {
        let mut eng = ENGINE.lock().unwrap();
        if !eng.available() { return input_data.clone(); }
        eng.record_tilt(true, self.tilt_x, self.tilt_y);
        let mut output = shallow_clone(input_data);
        // the extended axis, so that the stage loop's re-plan check (data_thread.rs:1194-1227) sees the new length
        let t = input_data.time.as_slice().unwrap();
        let (dx, dy) = (input_data.dx.unwrap_or(1.0), input_data.dy.unwrap_or(1.0));
        let steps = unsafe { thz_host_tilt_plan(t.as_ptr(), t.len(), input_data.width, input_data.height, self.tilt_x, self.tilt_y,
                                                dx, dy, std::ptr::null_mut(), std::ptr::null_mut()) };
        if steps > 0 {
            let mut new_time = vec![0f32; t.len() + 2 * steps];
            unsafe { thz_host_tilt_plan(t.as_ptr(), t.len(), input_data.width, input_data.height, self.tilt_x, self.tilt_y,
                                        dx, dy, new_time.as_mut_ptr(), std::ptr::null_mut()); }
            output.time = Array1::from(new_time);
        }
        output
    }

    fn ui(&mut self, ui: &mut Ui, _thread_communication: &mut ThreadCommunication, _panel_width: f32) -> egui::Response {
        let mut final_response = ui.allocate_response(egui::Vec2::ZERO, egui::Sense::hover());
        let rx = ui.horizontal(|ui| { ui.label("Tilt X: "); ui.add(egui::Slider::new(&mut self.tilt_x, -15.0..=15.0).suffix(" deg")) }).inner;
        let ry = ui.horizontal(|ui| { ui.label("Tilt Y: "); ui.add(egui::Slider::new(&mut self.tilt_y, -15.0..=15.0).suffix(" deg")) }).inner;
        final_response |= rx.clone();
        final_response |= ry.clone();
        if rx.changed() || ry.changed() { final_response.mark_changed(); }
        final_response
    }
}
