//! "Time Band Pass" (before the FFT) on the engine — REPLACES ONLY the bodies of `filter()` and `show_data()` in
//! `src/filters/band_pass_td_before_fft.rs` (`:124-182`, `:79-100`).  The struct, `new`, `reset`, `config`, `ui`
//! (the double slider over `time_axis` with the `signal_axis` plot) and the unit tests stay VERBATIM.
//! Transliteration of `GpuTimeDomainBandPassBeforeFFT` (`thz_image_explorer_amd/host/thz_engine.cpp`, tested by
//! `tests/test_gpu_engine.py`); UNVERIFIED BY A COMPILER.
//!
//! Add to the file's imports:
//!     use crate::gpu::engine::ENGINE;
//!     use crate::gpu::ffi::{thz_host_adapted_blackman, thz_host_td_bandpass};
//!     use crate::math_tools_gpu::{empty_plot_out, shallow_clone};

    /// Called by the data thread AFTER the recompute (deferred, `rust/data_thread.patch`).  The reference plots the
    /// stage's own output — the selected pixel's trace behind the Tilt taper and this band pass.  The engine keeps no
    /// per-stage cubes; for one trace the stage is two host multiplies on the raw trace (a tilted or scaled chain
    /// shows the raw trace itself: its stage output sits on another axis / grid).
    fn show_data(&mut self, data: &ScannedImageFilterData) {
        if data.width == 0 || data.height == 0 || data.time.is_empty() { return; }
        let eng = ENGINE.lock().unwrap();
        if !eng.available() { drop(eng); return self.show_data_cpu(data); }   // the reference's body, kept as `show_data_cpu`
        self.time_axis = data.time.to_vec();
        let mut v = vec![0f32; eng.nt];
        let mut po = empty_plot_out();
        po.signal = v.as_mut_ptr();
        if !eng.plot(data.pixel_selected[0] * data.scaling, data.pixel_selected[1] * data.scaling, &po) { return; }
        if v.len() == data.time.len() && data.scaling <= 1 {
            let t = data.time.as_slice().unwrap();
            let mut w = vec![0f32; v.len()];
            if eng.pending.tilt_active != 0 {
                unsafe { thz_host_adapted_blackman(t.as_ptr(), t.len(), 0.0, 7.0, w.as_mut_ptr()); }
                for (x, m) in v.iter_mut().zip(&w) { *x *= m; }
            }
            let (mut lo, mut hi) = (self.low, self.high);
            unsafe { thz_host_td_bandpass(t.as_ptr(), t.len(), &mut lo, &mut hi, self.window_width, w.as_mut_ptr(), std::ptr::null_mut(), std::ptr::null_mut()); }
            for (x, m) in v.iter_mut().zip(&w) { *x *= m; }
        }
        self.signal_axis = v.clone();
        self.input_signal_axis = v;
    }

    fn filter(&mut self, input_data: &ScannedImageFilterData, gui_settings: &mut GuiSettingsContainer,
              progress_lock: &mut Arc<RwLock<Option<f32>>>, abort_flag: &Arc<AtomicBool>) -> ScannedImageFilterData {
        let mut eng = ENGINE.lock().unwrap();
        if !eng.available() { drop(eng); return self.filter_cpu(input_data, gui_settings, progress_lock, abort_flag); }
        // clamps self.low / self.high like :137-138 (the index rule itself runs in the engine: thz_host_td_bandpass)
        let t = input_data.time.as_slice().unwrap();
        let mut w = vec![0f32; t.len()];
        unsafe { thz_host_td_bandpass(t.as_ptr(), t.len(), &mut self.low, &mut self.high, self.window_width, w.as_mut_ptr(), std::ptr::null_mut(), std::ptr::null_mut()); }
        eng.record_td_before(true, self.low, self.high, self.window_width);
        shallow_clone(input_data)
    }
