//! "Time Band Pass" (after the inverse FFT) on the engine — REPLACES ONLY the bodies of `filter()` and `show_data()`
//! in `src/filters/band_pass_td_after_fft.rs`.  The struct, `new`, `reset`, `config`, `ui` and the unit tests stay
//! VERBATIM.  Transliteration of `GpuTimeDomainBandPassAfterFFT` (`thz_image_explorer_amd/host/thz_engine.cpp`, tested
//! by `tests/test_gpu_engine.py`); UNVERIFIED BY A COMPILER.
//!
//! Add to the file's imports:
//!     use crate::gpu::engine::ENGINE;
//!     use crate::gpu::ffi::thz_host_td_bandpass;
//!     use crate::math_tools_gpu::{empty_plot_out, shallow_clone};

    /// Called AFTER the recompute (deferred): the selected pixel's trace of this stage's output = the chain's final
    /// trace (this is the last stage in front of the Deconvolution)
    fn show_data(&mut self, data: &ScannedImageFilterData) {
        if data.width == 0 || data.height == 0 || data.time.is_empty() { return; }
        let eng = ENGINE.lock().unwrap();
        if !eng.available() { drop(eng); return self.show_data_cpu(data); }
        self.time_axis = data.time.to_vec();
        let mut v = vec![0f32; eng.nt_out()];
        let mut po = empty_plot_out();
        po.filtered_signal = v.as_mut_ptr();
        if eng.plot(data.pixel_selected[0] * data.scaling, data.pixel_selected[1] * data.scaling, &po) {
            self.signal_axis = v.clone();
            self.input_signal_axis = v;
        }
    }

    fn filter(&mut self, input_data: &ScannedImageFilterData, gui_settings: &mut GuiSettingsContainer,
              progress_lock: &mut Arc<RwLock<Option<f32>>>, abort_flag: &Arc<AtomicBool>) -> ScannedImageFilterData {
        let mut eng = ENGINE.lock().unwrap();
        if !eng.available() { drop(eng); return self.filter_cpu(input_data, gui_settings, progress_lock, abort_flag); }
        let t = input_data.time.as_slice().unwrap();
        let mut w = vec![0f32; t.len()];
        unsafe { thz_host_td_bandpass(t.as_ptr(), t.len(), &mut self.low, &mut self.high, self.window_width, w.as_mut_ptr(), std::ptr::null_mut(), std::ptr::null_mut()); }
        eng.record_td_after(true, self.low, self.high, self.window_width);
        shallow_clone(input_data)
    }
