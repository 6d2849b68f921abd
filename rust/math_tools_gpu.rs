//! `src/math_tools_gpu.rs` — `math_tools::{scaling, fft, ifft}` on the engine, and the few helpers
//! `rust/data_thread.patch` calls from the data thread.
//!
//! TRANSLITERATION of `thz_image_explorer_amd/host/thz_engine.cpp` (`namespace math_tools_gpu`, `GpuPipeline::
//! update_filter`), which is built and tested (`tests/test_gpu_engine.py`); UNVERIFIED BY A COMPILER here.
//!
//! The three stage functions keep the reference's signatures (`src/math_tools.rs:242, 330, 418`): the bodies of
//! `math_tools::{scaling, fft, ifft}` become `if engine_available() { math_tools_gpu::x(input, config) } else
//! { <the reference's body> }`, so the call sites `data_thread.rs:1109-1120` stay as they are.  Each call records its
//! parameters and returns the container's metadata and axes; the arrays stay on the device until `finish_stage_walk`.
//! Without an engine (no GPU, no `libthzgpu.so`) the reference's own code runs: the engine has no CPU compute path.
use crate::config::{ConfigContainer, ThreadCommunication};
use crate::data_container::{PlotDataContainer, ScannedImageFilterData};
use crate::filters::filter::{Filter, FilterConfig};
use crate::gpu::engine::{chain_position_of_domain, chain_position_of_id, Polygon, ENGINE};
use crate::gpu::ffi::{ThzPlotOut, ThzRoiOut, ThzVoxelCfg};
use ndarray::{Array1, Array2, Array3};
use num_complex::Complex32;
use std::ptr;

pub fn engine_available() -> bool { ENGINE.lock().map(|e| e.available()).unwrap_or(false) }

/// metadata, axes, regions and plans of `input`; the five big arrays are left empty (shape 0): they are resident on
/// the device
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

/// `ConfigCommand::OpenFile`, behind `open_scan_from_thz` (`data_thread.rs:598-722`): the loader has subtracted the
/// bias and summed the first image (`io.rs:578-596`); the cube goes to the device(s) as it is.  Slot 0 keeps its host
/// copy (the plot's raw trace, the metadata editor); -> whether the engine took the scan.
pub fn open_scan(scan: &ScannedImageFilterData) -> bool {
    let mut eng = ENGINE.lock().unwrap();
    if !eng.available() { return false; }
    let (nx, ny, _nt) = scan.data.dim();
    match (scan.data.as_slice(), scan.time.as_slice()) {
        (Some(c), Some(t)) => eng.open_scan(c, nx, ny, t, scan.dx.unwrap_or(1.0), scan.dy.unwrap_or(1.0), false),
        _ => false,
    }
}

/// chain position 1 (`data_thread.rs:1109-1112`); metadata exactly as `math_tools.rs:250-270`
pub fn scaling(input: &ScannedImageFilterData, config: &ConfigContainer) -> ScannedImageFilterData {
    let s = config.scale_factor;
    ENGINE.lock().unwrap().record_scaling(s);
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
    if input.r2c.is_none() { return input.clone(); }               // math_tools.rs:332
    ENGINE.lock().unwrap().record_fft(config.fft_window_type as i32, config.fft_window[0], config.fft_window[1]);
    shallow_clone(input)
}

/// chain position 6 (`data_thread.rs:1117-1120`): pixel means, per-region means, C2R / nt — the stage's regions
/// (`input.rois`, `math_tools.rs:473-475`: those with a polygon) and `avg_in_fourier_space` go with it
pub fn ifft(input: &ScannedImageFilterData, config: &ConfigContainer) -> ScannedImageFilterData {
    let mut rois: Vec<(String, Polygon)> = input.rois.iter()
        .filter_map(|(uuid, (_name, poly))| poly.as_ref().map(|p| (uuid.clone(), p.clone())))
        .collect();
    rois.sort_by(|a, b| a.0.cmp(&b.0));   // HashMap order is arbitrary; the device keeps them in this order
    ENGINE.lock().unwrap().record_ifft(config.avg_in_fourier_space, rois);
    shallow_clone(input)
}

/// Top of the stage loop (`data_thread.rs:1090`): tells the engine at which of ITS chain positions the walk starts
/// (the reference's chain has one entry per Frequency plugin, the engine one position for all of them) -> whether
/// the engine is in use for this walk.
pub fn begin_walk(filter_chain: &[String], start_idx: usize, filters: &[(String, Box<dyn Filter>)]) -> bool {
    let mut eng = ENGINE.lock().unwrap();
    if !eng.available() { return false; }
    let first = &filter_chain[start_idx.min(filter_chain.len() - 1)];
    let mut pos = chain_position_of_id(first);
    if pos == 0 {
        if let Some((_, f)) = filters.iter().find(|(id, _)| id == first) { pos = chain_position_of_domain(&f.config().domain); }
    }
    eng.begin_walk(pos.max(1));
    true
}

/// the walk passed this (inactive) plugin's input through (`data_thread.rs:1185-1188`)
pub fn note_inactive(cfg: &FilterConfig) { ENGINE.lock().unwrap().note_inactive(cfg); }

fn expand_image(img: &[f32], gx: usize, gy: usize, s: usize) -> Array2<f32> {
    // data_thread.rs:1243-1285: the value of a block fills its s x s pixels
    let mut big = Array2::<f32>::zeros((gx * s, gy * s));
    for x in 0..gx { for y in 0..gy { for a in 0..s { for b in 0..s { big[[x * s + a, y * s + b]] = img[x * gy + y]; } } } }
    big
}

/// `UpdateType::Image`: the image of the current results (`data_thread.rs:1562-1652`)
pub fn refresh_image(last: &mut ScannedImageFilterData) {
    let eng = ENGINE.lock().unwrap();
    if let Some((img, gx, gy)) = eng.image() {
        let s = last.scaling.max(1);
        last.img = if s > 1 { expand_image(&img, gx, gy, s) } else { Array2::from_shape_vec((gx, gy), img).unwrap() };
    }
}

/// Where the reference sums the image (`data_thread.rs:1242-1308`): ONE recompute from the lowest chain position the
/// walk touched, then everything the code behind the loop reads from the LAST container: `img`, the three averages,
/// `avg_data`, and the regions' maps (`roi_signal_fft` / `roi_phase_fft` / `roi_data`: the optical-properties block
/// reads them, `:1490-1556`).
pub fn finish_stage_walk(last: &mut ScannedImageFilterData, config: &ConfigContainer) -> bool {
    let mut eng = ENGINE.lock().unwrap();
    if !eng.flush() {
        log::error!("gpu recompute failed ({}): results of the previous run are kept", eng.last_error());
        return false;
    }
    let nto = eng.nt_out();
    let nf = nto / 2 + 1;
    if let Some((img, gx, gy)) = eng.image() {
        let s = last.scaling.max(1);
        last.img = if s > 1 { expand_image(&img, gx, gy, s) } else { Array2::from_shape_vec((gx, gy), img).unwrap() };
    }
    if let Some((f, a, p)) = eng.averages() {                       // math_tools.rs:421-440
        last.avg_fft = Array1::from(f);
        last.avg_signal_fft = Array1::from(a);
        last.avg_phase_fft = Array1::from(p);
    }
    let (px, py) = (last.pixel_selected[0] * last.scaling, last.pixel_selected[1] * last.scaling);
    if config.avg_in_fourier_space {                                // math_tools.rs:442-470
        let mut avg = vec![0f32; nto];
        let mut po = empty_plot_out();
        po.avg_signal = avg.as_mut_ptr();
        if eng.plot(px, py, &po) { last.avg_data = Array1::from(avg); }
    }
    last.roi_signal_fft.clear();
    last.roi_phase_fft.clear();
    last.roi_data.clear();
    for (uuid, (name, poly)) in last.rois.iter() {                  // math_tools.rs:473-543
        if poly.is_none() { continue; }
        let (mut a, mut p, mut d) = (vec![0f32; nf], vec![0f32; nf], vec![0f32; nto]);
        let ro = ThzRoiOut { signal_fft: a.as_mut_ptr(), phase_fft: p.as_mut_ptr(), signal: ptr::null_mut(), roi_data: d.as_mut_ptr(), count: ptr::null_mut() };
        if !eng.roi(uuid, &ro) { continue; }
        last.roi_signal_fft.insert(uuid.clone(), (name.clone(), Array1::from(a)));
        last.roi_phase_fft.insert(uuid.clone(), (name.clone(), Array1::from(p)));
        last.roi_data.insert(uuid.clone(), (name.clone(), Array1::from(d)));
    }
    true
}

pub fn empty_plot_out() -> ThzPlotOut {
    ThzPlotOut { signal: ptr::null_mut(), signal_fft: ptr::null_mut(), phase_fft: ptr::null_mut(), filtered_signal: ptr::null_mut(),
                 filtered_signal_fft: ptr::null_mut(), filtered_phase_fft: ptr::null_mut(), avg_signal: ptr::null_mut(),
                 avg_signal_fft: ptr::null_mut(), avg_phase_fft: ptr::null_mut() }
}

/// The plot copy-out (`data_thread.rs:1337-1488`, and again `:1655-1760`): selected pixel, averages and every
/// region's vectors in `PlotDataContainer`, from the device.
pub fn fill_plot_data(data: &mut PlotDataContainer, filter_data: &[ScannedImageFilterData], _config: &ConfigContainer, fft_index: usize) {
    let eng = ENGINE.lock().unwrap();
    let (raw, filtered) = match (filter_data.first(), filter_data.last()) { (Some(r), Some(f)) => (r, f), _ => return };
    let nto = eng.nt_out();
    let nf = nto / 2 + 1;
    // pixel of the RAW grid: slot 0 is never scaled; the engine divides once for everything behind the scaling stage
    let (px, py) = (raw.pixel_selected[0], raw.pixel_selected[1]);
    if px >= eng.nx || py >= eng.ny {
        log::warn!("selected pixel ({px}, {py}) is out of bounds for the scan ({} x {})", eng.nx, eng.ny);
        return;
    }
    let (mut signal, mut sig_fft, mut ph_fft) = (vec![0f32; eng.nt], vec![0f32; nf], vec![0f32; nf]);
    let (mut f_sig, mut f_fft, mut f_ph) = (vec![0f32; nto], vec![0f32; nf], vec![0f32; nf]);
    let (mut avg, mut avg_a, mut avg_p) = (vec![0f32; nto], vec![0f32; nf], vec![0f32; nf]);
    let po = ThzPlotOut { signal: signal.as_mut_ptr(), signal_fft: sig_fft.as_mut_ptr(), phase_fft: ph_fft.as_mut_ptr(),
                          filtered_signal: f_sig.as_mut_ptr(), filtered_signal_fft: f_fft.as_mut_ptr(), filtered_phase_fft: f_ph.as_mut_ptr(),
                          avg_signal: avg.as_mut_ptr(), avg_signal_fft: avg_a.as_mut_ptr(), avg_phase_fft: avg_p.as_mut_ptr() };
    if !eng.plot(px, py, &po) { log::error!("gpu plot copy-out: {}", eng.last_error()); return; }
    data.time = raw.time.to_vec();
    data.signal = signal;
    if let Some(spec) = filter_data.get(fft_index + 1) { data.frequencies = spec.frequency.to_vec(); }
    data.signal_fft = sig_fft;
    data.phase_fft = ph_fft;
    data.filtered_time = filtered.time.to_vec();
    data.filtered_signal = f_sig;
    data.filtered_frequencies = filtered.frequency.to_vec();
    data.filtered_signal_fft = f_fft;
    data.filtered_phase_fft = f_ph;
    data.avg_signal = avg;          // the final cube's pixel mean, or avg_data with avg_in_fourier_space (:1422-1432)
    data.avg_signal_fft = avg_a;
    data.avg_phase_fft = avg_p;
    for (uuid, (name, poly)) in filtered.rois.iter() {               // :1442-1482
        if poly.is_none() { continue; }
        let (mut s, mut a, mut p) = (vec![0f32; nto], vec![0f32; nf], vec![0f32; nf]);
        // `signal`: mean of the final traces, or roi_data when averaging in Fourier space — the engine picks (:1476-1482)
        let ro = ThzRoiOut { signal_fft: a.as_mut_ptr(), phase_fft: p.as_mut_ptr(), signal: s.as_mut_ptr(), roi_data: ptr::null_mut(), count: ptr::null_mut() };
        if !eng.roi(uuid, &ro) { continue; }
        data.roi_signal.insert(uuid.to_string(), (name.clone(), s));
        data.roi_signal_fft.insert(uuid.to_string(), (name.clone(), a));
        data.roi_phase.insert(uuid.to_string(), (name.clone(), p));
    }
}

/// amplitudes and phases of the selected pixel for the optical-properties block (`data_thread.rs:1497-1513`); `None`
/// without the engine (the reference then indexes its host arrays)
pub fn selected_pixel_spectrum(filtered: &ScannedImageFilterData) -> Option<(Array1<f32>, Array1<f32>)> {
    let eng = ENGINE.lock().unwrap();
    if !eng.available() { return None; }
    let nf = eng.nt_out() / 2 + 1;
    let (mut a, mut p) = (vec![0f32; nf], vec![0f32; nf]);
    let mut po = empty_plot_out();
    po.filtered_signal_fft = a.as_mut_ptr();
    po.filtered_phase_fft = p.as_mut_ptr();
    if !eng.plot(filtered.pixel_selected[0] * filtered.scaling, filtered.pixel_selected[1] * filtered.scaling, &po) { return None; }
    Some((Array1::from(a), Array1::from(p)))
}

/// `update_intensity_image`'s 3-D part (`data_thread.rs:64-101`, `gui/threed_plot.rs:132-276`) on the device.  The
/// envelope parameters are the GUI's (`ThreadCommunication::gui_settings`: sigma, radius, contrast, opacity threshold)
pub fn voxel_instances(_time_span: f32, scaling: usize, original_dims: (usize, usize, usize), tc: &ThreadCommunication)
                       -> (Vec<bevy_voxel_plot::InstanceData>, f32, f32, f32) {
    let eng = ENGINE.lock().unwrap();
    let g = &tc.gui_settings;
    let cfg = ThzVoxelCfg { sigma: g.kernel_sigma, radius: g.kernel_radius as i32, contrast: g.contrast_3d, opacity_threshold: g.opacity_threshold };
    match eng.voxels(&cfg, 2_000_000, scaling, original_dims) {
        Some((inst, thr, dims)) => {
            if let Ok(mut t) = tc.opacity_threshold_lock.write() { *t = thr; }
            // `ThzVoxelInstance` has the layout `instance_from_data` fills `InstanceData` with (threed_plot.rs:260-264)
            let inst = unsafe { std::mem::transmute::<Vec<crate::gpu::ffi::ThzVoxelInstance>, Vec<bevy_voxel_plot::InstanceData>>(inst) };
            (inst, dims[0], dims[1], dims[2])
        }
        None => (vec![], 1.0, 1.0, 1.0),
    }
}
