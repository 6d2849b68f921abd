//! "Tilt Compensation" on the engine — REPLACES ONLY the body of `filter()` in `src/filters/tilt_compensation.rs`
//! (`:97-226`).  The struct, `new`, `reset`, `show_data`, `config`, `ui` and the unit tests of that file stay
//! VERBATIM.  Transliteration of `GpuTiltCompensation::filter` (`thz_image_explorer_amd/host/thz_engine.cpp`, tested by
//! `tests/test_gpu_engine.py`); UNVERIFIED BY A COMPILER.
//!
//! Add to the file's imports:
//!     use crate::gpu::engine::ENGINE;
//!     use crate::gpu::ffi::{thz_host_frequency_axis, thz_host_tilt_plan};
//!     use crate::math_tools_gpu::shallow_clone;

    fn filter(&mut self, input_data: &ScannedImageFilterData, gui_settings: &mut GuiSettingsContainer,
              progress_lock: &mut Arc<RwLock<Option<f32>>>, abort_flag: &Arc<AtomicBool>) -> ScannedImageFilterData {
        let mut eng = ENGINE.lock().unwrap();
        if !eng.available() { drop(eng); return self.filter_cpu(input_data, gui_settings, progress_lock, abort_flag); } // the reference's body, kept as `filter_cpu`
        let (dx, dy) = match (input_data.dx, input_data.dy) { (Some(dx), Some(dy)) => (dx, dy), _ => return input_data.clone() }; // :111
        eng.record_tilt(true, self.tilt_x, self.tilt_y);
        let mut output = shallow_clone(input_data);
        // the extended axis (:104-170, 206-217): the containers behind this stage carry it; the device lays the cube
        // out on it with the same plan (thz_host_tilt_plan: extension, per-pixel insert index)
        let t = input_data.time.as_slice().unwrap();
        let steps = unsafe { thz_host_tilt_plan(t.as_ptr(), t.len(), input_data.width, input_data.height, self.tilt_x, self.tilt_y, dx, dy,
                                                std::ptr::null_mut(), std::ptr::null_mut()) };
        if steps > 0 {
            let mut new_time = vec![0f32; t.len() + 2 * steps];
            unsafe { thz_host_tilt_plan(t.as_ptr(), t.len(), input_data.width, input_data.height, self.tilt_x, self.tilt_y, dx, dy,
                                        new_time.as_mut_ptr(), std::ptr::null_mut()); }
            let mut freq = vec![0f32; new_time.len() / 2 + 1];
            unsafe { thz_host_frequency_axis(new_time.as_ptr(), new_time.len(), freq.as_mut_ptr()); }
            output.time = Array1::from(new_time);
            output.frequency = Array1::from(freq);
            // (the data thread re-plans r2c / c2r for the new length itself, data_thread.rs:1194-1227)
        }
        output
    }
