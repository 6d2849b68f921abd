//! `math_tools::{scaling, fft, ifft}` with the reference's signatures (`src/math_tools.rs:242, 330, 418`),
//! delegating to the engine.  Each call records its parameters (`GpuEngine::record_*`) and returns the
//! container's metadata and axes; the arrays stay on the device until `ENGINE.flush()` after the stage loop.
//! When no GPU engine is available the reference's own bodies (kept as `*_cpu`) run instead — the engine
//! itself has no CPU compute path.
//!
//! UNVERIFIED BY A COMPILER (no Rust toolchain in the authoring image).
use crate::config::ConfigContainer;
use crate::data_container::ScannedImageFilterData;
use crate::gpu::engine::ENGINE;
use ndarray::{Array1, Array2, Array3};
use num_complex::Complex32;

/// metadata, axes, ROIs and plans of `input`; the five big arrays are left empty (shape 0) — they are resident
/// on the device (`THZ_BUF_*`) and fetched by what needs them
pub fn shallow_clone(input: &ScannedImageFilterData) -> ScannedImageFilterData {
    ScannedImageFilterData {
        x_min: input.x_min, dx: input.dx, y_min: input.y_min, dy: input.dy,
        height: input.height, width: input.width, scaling: input.scaling, pixel_selected: input.pixel_selected,
        r2c: input.r2c.clone(), c2r: input.c2r.clone(), rois: input.rois.clone(),
        time: input.time.clone(), img: Array2::zeros((0, 0)), data: Array3::zeros((0, 0, 0)),
        avg_data: input.avg_data.clone(), datasets: input.datasets.clone(), roi_data: input.roi_data.clone(),
        frequency: input.frequency.clone(), fft: Array3::<Complex32>::zeros((0, 0, 0)),
        amplitudes: Array3::zeros((0, 0, 0)), phases: Array3::zeros((0, 0, 0)),
        avg_fft: input.avg_fft.clone(), avg_signal_fft: input.avg_signal_fft.clone(),
        avg_phase_fft: input.avg_phase_fft.clone(), roi_signal_fft: input.roi_signal_fft.clone(),
        roi_phase_fft: input.roi_phase_fft.clone(),
    }
}

/// chain position 1 (`data_thread.rs:1109-1112`); metadata exactly as `math_tools.rs:250-270`
pub fn scaling(input: &ScannedImageFilterData, config: &ConfigContainer) -> ScannedImageFilterData {
    let mut eng = ENGINE.lock().unwrap();
    if !eng.available() { return crate::math_tools::scaling_cpu(input, config); }
    let s = config.scale_factor;
    eng.record_scaling(s);
    let mut output = shallow_clone(input);
    if s > 1 && input.width / s > 0 && input.height / s > 0 {
        output.width = input.width / s;
        output.height = input.height / s;
        output.scaling = s;
        output.dx = output.dx.map(|d| d * s as f32);
        output.dy = output.dy.map(|d| d * s as f32);
        output.pixel_selected = [input.pixel_selected[0] / s, input.pixel_selected[1] / s];
    }
    output
}

/// chain position 4 (`data_thread.rs:1113-1116`): window + R2C + |.| + arg + numpy_unwrap
pub fn fft(input: &ScannedImageFilterData, config: &ConfigContainer) -> ScannedImageFilterData {
    let mut eng = ENGINE.lock().unwrap();
    if !eng.available() { return crate::math_tools::fft_cpu(input, config); }
    if input.r2c.is_none() { return input.clone(); }               // math_tools.rs:332
    eng.record_fft(config.fft_window_type as i32, config.fft_window[0], config.fft_window[1]);
    shallow_clone(input)
}

/// chain position 6 (`data_thread.rs:1117-1120`): pixel means, ROI means, C2R / nt
pub fn ifft(input: &ScannedImageFilterData, config: &ConfigContainer) -> ScannedImageFilterData {
    let mut eng = ENGINE.lock().unwrap();
    if !eng.available() { return crate::math_tools::ifft_cpu(input, config); }
    let _ = config;
    eng.record_ifft();
    shallow_clone(input)
}

/// Called once after the stage loop (`data_thread.rs:1229`): runs the recompute and fills what the code behind
/// it reads from the LAST container: `img` (`:1288-1307`) and the three averages (`math_tools.rs:421-440`).
pub fn finish_stage_walk(last: &mut ScannedImageFilterData) {
    let mut eng = ENGINE.lock().unwrap();
    if !eng.available() { return; }
    if let Err(e) = eng.flush() {
        log::error!("gpu recompute failed ({}: {}): results of the previous run are kept", e.0, e.1);
        return;
    }
    match eng.image() {
        Ok(img) => last.img = img,
        Err(e) => log::error!("gpu image download: {}", e.1),
    }
    if let Ok((f, a, p)) = eng.averages() {
        last.avg_fft = f;
        last.avg_signal_fft = a;
        last.avg_phase_fft = p;
    }
    let _ = Array1::<f32>::zeros(0);
}
