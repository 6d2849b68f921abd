//! "Deconvolution" on the engine — REPLACES ONLY the body of `filter()` in `src/filters/deconvolution.rs`
//! (`:766-1041`); add `psf_view` and its helpers next to it.  The struct, `new`, `reset`, `show_data`, `config`, `ui`
//! (iterations slider, expert mode with the bank parameters, `:1043-…`) and the unit tests stay VERBATIM.
//! Transliteration of `GpuDeconvolution::filter` (`thz_image_explorer_amd/host/thz_engine.cpp`, tested by
//! `tests/test_gpu_engine.py` on one slab and on a group of two); UNVERIFIED BY A COMPILER.
//!
//! The stage runs `thz_group_session_deconvolve` — FIR bank, per-band energy image, Richardson-Lucy against the
//! band's Gaussian PSF (all bands batched per iteration), gains, ONE recombination transform — over the WHOLE grid
//! whatever the number of GPUs.  The guards of the reference (empty PSF, image < 16 x 16, PSF wider than the image:
//! `:790-885`) come back as `THZ_SKIPPED`, an abort as `THZ_ERR_ABORTED`; in both cases the stage's input is what it
//! outputs, as in the reference.  Progress is written into `progress_lock` while the call runs and the ✕ button's
//! `abort_flag` is seen between iteration batches (`GpuEngine::deconvolve`).
//!
//! Add to the file's imports:
//!     use crate::gpu::engine::ENGINE;
//!     use crate::gpu::ffi::{ThzDeconvCfg, ThzHybridFit, ThzPsf, ThzSpline, THZ_SKIPPED};
//!     use crate::math_tools_gpu::shallow_clone;

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

    fn filter(&mut self, input_data: &ScannedImageFilterData, gui_settings: &mut GuiSettingsContainer,
              progress_lock: &mut Arc<RwLock<Option<f32>>>, abort_flag: &Arc<AtomicBool>) -> ScannedImageFilterData {
        let mut eng = ENGINE.lock().unwrap();
        if !eng.available() { drop(eng); return self.filter_cpu(input_data, gui_settings, progress_lock, abort_flag); }
        if input_data.dx.is_none() || input_data.dy.is_none() {           // :781
            log::error!("No data loaded, skipping deconvolution.");
            return input_data.clone();
        }
        let psf = psf_view(&gui_settings.psf);
        let cfg = ThzDeconvCfg { n_iterations: self.n_iterations as u32, n_filters: self.n_filters as u32, start_freq: self.start_freq,
                                 end_freq: self.end_freq, win_width: self.win_width, band_begin: 0, band_end: 0 };
        // flushes everything in front of the stage first, then runs it over the whole group
        let rc = eng.deconvolve(&psf, &cfg, progress_lock, abort_flag);
        if rc == THZ_SKIPPED { log::warn!("Deconvolution: a guard of the reference applied, input returned unchanged"); }
        else if rc < 0 { log::error!("Deconvolution failed or was aborted ({}), the stage passes its input through", eng.last_error()); }
        shallow_clone(input_data)
    }
