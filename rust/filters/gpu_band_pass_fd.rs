//! "Frequency Band Pass" on the engine — REPLACES ONLY the bodies of `filter()` and `show_data()` in
//! `src/filters/band_pass_fd.rs` (`:122-220`, `:74-97`).  The struct, `new`, `reset`, `config`, `ui` (the double
//! slider over `freq_axis` with the `signal_axis` spectrum plot) and the unit tests stay VERBATIM.
//! Transliteration of `GpuFrequencyDomainBandPass` (`thz_image_explorer_amd/host/thz_engine.cpp`, tested by
//! `tests/test_gpu_engine.py`); UNVERIFIED BY A COMPILER.
//!
//! Add to the file's imports:
//!     use crate::gpu::engine::ENGINE;
//!     use crate::math_tools_gpu::{empty_plot_out, shallow_clone};

    /// Called AFTER the recompute (deferred): |band-passed spectrum| of the selected pixel — the reference takes
    /// `norm()` of `data.fft[pixel]`, which is the band-passed amplitude the device stores
    fn show_data(&mut self, data: &ScannedImageFilterData) {
        if data.width == 0 || data.height == 0 || data.frequency.is_empty() { return; }
        let eng = ENGINE.lock().unwrap();
        if !eng.available() { drop(eng); return self.show_data_cpu(data); }
        self.freq_axis = data.frequency.to_vec();
        let mut amp = vec![0f32; eng.nt_out() / 2 + 1];
        let mut po = empty_plot_out();
        po.filtered_signal_fft = amp.as_mut_ptr();
        if eng.plot(data.pixel_selected[0] * data.scaling, data.pixel_selected[1] * data.scaling, &po) { self.signal_axis = amp; }
    }

    fn filter(&mut self, input_data: &ScannedImageFilterData, gui_settings: &mut GuiSettingsContainer,
              progress_lock: &mut Arc<RwLock<Option<f32>>>, abort_flag: &Arc<AtomicBool>) -> ScannedImageFilterData {
        let mut eng = ENGINE.lock().unwrap();
        if !eng.available() { drop(eng); return self.filter_cpu(input_data, gui_settings, progress_lock, abort_flag); }
        // index rule :135-168 and zero padding :194-212 run in the engine (thz_host_fd_bandpass), fused into the launch
        eng.record_fd(true, self.low, self.high, self.window_width);
        shallow_clone(input_data)
    }
