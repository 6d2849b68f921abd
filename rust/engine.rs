//! Process-global record-and-flush engine behind the data thread (`src/gpu/engine.rs`).
//!
//! TRANSLITERATION of `thz_image_explorer_amd/host/thz_engine.{hpp,cpp}` (class `GpuEngine`): same fields, same
//! methods, same order of calls.  No Rust toolchain exists in the image this repository is built in, so the logic
//! is written, built and tested in C++ first (`tests/test_gpu_engine.py` drives it through the patched stage walk
//! against the oracle); this file is that code in the reference's language and is UNVERIFIED BY A COMPILER.
//!
//! Why an engine object at all: `Filter` objects are cloned before every run and only `#[static_field]` members are
//! copied back (`data_thread.rs:1069-1078, 1322-1334`), so device state cannot live in a plugin struct; it lives
//! here, behind a `once_cell::Lazy` (`once_cell` is already a dependency, `Cargo.toml:69`).  One `thz_group` drives
//! every GPU of the node from the single data thread (`data_thread.rs:162-174`); with one GPU the group is trivial
//! and RCCL is never loaded.
//!
//! How it meets the stage walk (`data_thread.rs:1090-1191`, as `rust/data_thread.patch` leaves it): the device
//! computes the whole chain in one launch, so a stage call only RECORDS its parameters — or, for a plugin the walk
//! passes through, its inactivity (`note_inactive`) — and returns a container without the big arrays; behind the
//! loop one `flush()` runs `thz_group_session_recompute` from the lowest chain position the walk touched, and
//! everything the code behind the loop reads is fetched from the device then.
use super::ffi::*;
use crate::filters::filter::{FilterConfig, FilterDomain};
use num_complex::Complex32;
use once_cell::sync::Lazy;
use std::ffi::CStr;
use std::os::raw::{c_int, c_void};
use std::ptr;
use std::sync::atomic::{AtomicBool, AtomicI32, Ordering};
use std::sync::{Arc, Mutex, RwLock};

/// chain positions of `thz_session_recompute_from`: the reference's `filter_chain` (`main.rs:182-247`) with all
/// `FilterDomain::Frequency` plugins on one position
pub const POS_SCALING: usize = 1;
pub const POS_TILT: usize = 2;
pub const POS_TD_BEFORE: usize = 3;
pub const POS_FFT: usize = 4;
pub const POS_FREQUENCY: usize = 5;
pub const POS_IFFT: usize = 6;
pub const POS_TD_AFTER: usize = 7;
pub const POS_DECONVOLUTION: usize = 8;

pub fn chain_position_of_domain(d: &FilterDomain) -> usize {
    match d {
        FilterDomain::TimeBeforeFFTPrioFirst => POS_TILT,
        FilterDomain::TimeBeforeFFT => POS_TD_BEFORE,
        FilterDomain::Frequency => POS_FREQUENCY,
        FilterDomain::TimeAfterFFT => POS_TD_AFTER,
        FilterDomain::TimeAfterFFTPrioLast => POS_DECONVOLUTION,
    }
}
/// "scaling" / "fft" / "ifft", else 0
pub fn chain_position_of_id(id: &str) -> usize {
    match id {
        "scaling" => POS_SCALING,
        "fft" => POS_FFT,
        "ifft" => POS_IFFT,
        _ => 0,
    }
}

/// `THZGPU_DEVICES=0,1,2,3` selects the GPUs; default: device 0.
fn devices_from_env() -> Vec<c_int> {
    std::env::var("THZGPU_DEVICES")
        .ok()
        .map(|s| s.split(',').filter_map(|t| t.trim().parse().ok()).collect::<Vec<c_int>>())
        .filter(|v| !v.is_empty())
        .unwrap_or_else(|| vec![0])
}

pub static ENGINE: Lazy<Mutex<GpuEngine>> = Lazy::new(|| Mutex::new(GpuEngine::new(&devices_from_env())));

pub type Polygon = Vec<(usize, usize)>;

pub struct GpuEngine {
    group: *mut ThzGroup,
    session: *mut ThzGroupSession,
    /// what the stage calls of the current walk have recorded
    pub pending: ThzChainCfg,
    /// lowest chain position touched since the last flush: the `start_stage` of the recompute
    dirty_from: usize,
    pub nx: usize,
    pub ny: usize,
    pub nt: usize,
    /// K14 / K13 per-bin multipliers; empty = that plugin is off
    fd_real: Vec<f32>,
    fd_cmask: Vec<f32>,
    plugins_dirty: bool,
    /// (uuid, polygon) of the ifft stage's regions, in the order the device holds them
    rois: Vec<(String, Polygon)>,
    rois_dirty: bool,
}

// the raw handles are only touched under the Mutex around the engine
unsafe impl Send for GpuEngine {}

impl GpuEngine {
    pub fn new(devices: &[c_int]) -> Self {
        let mut group = ptr::null_mut();
        let rc = unsafe { thz_group_create(devices.as_ptr(), devices.len() as c_int, &mut group) };
        if rc != THZ_OK {
            // no CPU compute path exists in the engine: the caller keeps the reference's own path
            let why = unsafe { CStr::from_ptr(thz_group_last_error(ptr::null())) }.to_string_lossy().into_owned();
            log::error!("thz_group_create({devices:?}) failed ({why}): GPU path disabled");
            group = ptr::null_mut();
        }
        let mut pending: ThzChainCfg = unsafe { std::mem::zeroed() };
        pending.scale_factor = 1;
        pending.want_means = 1;
        GpuEngine { group, session: ptr::null_mut(), pending, dirty_from: 1, nx: 0, ny: 0, nt: 0, fd_real: vec![], fd_cmask: vec![],
                    plugins_dirty: true, rois: vec![], rois_dirty: false }
    }

    pub fn available(&self) -> bool { !self.group.is_null() }

    pub fn last_error(&self) -> String {
        if self.group.is_null() { return "no GPU group".into(); }
        unsafe { CStr::from_ptr(thz_group_last_error(self.group)) }.to_string_lossy().into_owned()
    }

    /// `ConfigCommand::OpenFile` (`io.rs:576-628`): the cube goes to the device(s) once.  `subtract_bias`: false when
    /// the loader already did it (`open_scan_from_thz` does, `io.rs:578-586`).
    pub fn open_scan(&mut self, cube: &[f32], nx: usize, ny: usize, time: &[f32], dx: f32, dy: f32, subtract_bias: bool) -> bool {
        if self.group.is_null() { return false; }
        unsafe {
            if !self.session.is_null() { thz_group_session_destroy(self.session); self.session = ptr::null_mut(); }
            if thz_group_session_create(self.group, nx, ny, time.len(), time.as_ptr(), dx, dy, &mut self.session) != THZ_OK
                || thz_group_session_upload(self.session, cube.as_ptr(), subtract_bias as c_int) != THZ_OK
            {
                log::error!("open_scan: {}", self.last_error());
                if !self.session.is_null() { thz_group_session_destroy(self.session); }
                self.session = ptr::null_mut();
                return false;
            }
            thz_chain_cfg_default(time.as_ptr(), time.len(), &mut self.pending);
        }
        self.nx = nx; self.ny = ny; self.nt = time.len();
        self.dirty_from = 1;
        self.fd_real.clear();
        self.fd_cmask.clear();
        self.plugins_dirty = true;
        self.rois.clear();
        self.rois_dirty = false;
        true
    }

    fn touch(&mut self, position: usize) { if position < self.dirty_from { self.dirty_from = position.max(1); } }

    // ---- the stage walk of UpdateType::Filter(start_idx)
    pub fn begin_walk(&mut self, start_position: usize) { self.touch(start_position); }
    pub fn record_scaling(&mut self, scale_factor: usize) { self.pending.scale_factor = scale_factor as i32; self.touch(POS_SCALING); }
    pub fn record_tilt(&mut self, active: bool, tilt_x: f64, tilt_y: f64) {
        self.pending.tilt_active = active as i32;
        if active { self.pending.tilt_x_deg = tilt_x; self.pending.tilt_y_deg = tilt_y; }
        self.touch(POS_TILT);
    }
    pub fn record_td_before(&mut self, active: bool, low: f64, high: f64, width: f64) {
        self.pending.td_before_active = active as i32;
        if active { self.pending.td_before_low = low; self.pending.td_before_high = high; self.pending.td_before_width = width; }
        self.touch(POS_TD_BEFORE);
    }
    pub fn record_fft(&mut self, window_type: i32, lower: f32, upper: f32) {
        self.pending.fft_window = ThzWindowCfg { type_: window_type, lower, upper };
        self.touch(POS_FFT);
    }
    pub fn record_fd(&mut self, active: bool, low: f64, high: f64, width: f64) {
        self.pending.fd_active = active as i32;
        if active { self.pending.fd_low = low; self.pending.fd_high = high; self.pending.fd_width = width; }
        self.touch(POS_FREQUENCY);
    }
    /// K14: real per-bin multiplier (nf).  An unchanged multiplier does not invalidate the resident spectrum.
    pub fn record_water_lines(&mut self, active: bool, mask: Vec<f32>) {
        let mask = if active { mask } else { vec![] };
        if mask != self.fd_real { self.fd_real = mask; self.plugins_dirty = true; }
        self.touch(POS_FREQUENCY);
    }
    /// K13: complex per-bin multiplier (2 nf, interleaved)
    pub fn record_wiener(&mut self, active: bool, cmask: Vec<f32>) {
        let cmask = if active { cmask } else { vec![] };
        if cmask != self.fd_cmask { self.fd_cmask = cmask; self.plugins_dirty = true; }
        self.touch(POS_FREQUENCY);
    }
    /// the ifft stage: its regions (`input.rois`, `math_tools.rs:473-475`) and `config.avg_in_fourier_space` go with it
    pub fn record_ifft(&mut self, avg_in_fourier_space: bool, rois: Vec<(String, Polygon)>) {
        self.pending.avg_in_fourier_space = avg_in_fourier_space as i32;
        if rois != self.rois { self.rois = rois; self.rois_dirty = true; }
        self.touch(POS_IFFT);
    }
    pub fn record_td_after(&mut self, active: bool, low: f64, high: f64, width: f64) {
        self.pending.td_after_active = active as i32;
        if active { self.pending.td_after_low = low; self.pending.td_after_high = high; self.pending.td_after_width = width; }
        self.touch(POS_TD_AFTER);
    }
    /// the walk passed an inactive plugin's input through (`data_thread.rs:1185-1188`): its stage is off in the chain
    pub fn note_inactive(&mut self, cfg: &FilterConfig) {
        match cfg.domain {
            FilterDomain::TimeBeforeFFTPrioFirst => self.record_tilt(false, 0.0, 0.0),
            FilterDomain::TimeBeforeFFT => self.record_td_before(false, 0.0, 0.0, 0.0),
            FilterDomain::Frequency => {
                if cfg.name == "Water Line Notch" { self.record_water_lines(false, vec![]) }
                else if cfg.name == "Reference Wiener Filter" { self.record_wiener(false, vec![]) }
                else { self.record_fd(false, 0.0, 0.0, 0.0) }
            }
            FilterDomain::TimeAfterFFT => self.record_td_after(false, 0.0, 0.0, 0.0),
            // the stage hands its input on: an earlier deconvolved cube must not stay the chain's output.  The tail of
            // the chain (C2R, Time Band Pass, image) is what restores the stage's input as the final cube.
            FilterDomain::TimeAfterFFTPrioLast => self.touch(POS_TD_AFTER),
        }
    }

    /// Behind the stage loop (and in front of the Deconvolution stage): one recompute from the lowest touched position.
    pub fn flush(&mut self) -> bool {
        if self.session.is_null() { return false; }
        unsafe {
            if self.plugins_dirty {
                for i in 0..thz_group_local_count(self.group) {
                    let s = thz_group_session_member(self.session, i);
                    let nf = if !self.fd_real.is_empty() { self.fd_real.len() } else { self.fd_cmask.len() / 2 };
                    let r = if self.fd_real.is_empty() { ptr::null() } else { self.fd_real.as_ptr() };
                    let c = if self.fd_cmask.is_empty() { ptr::null() } else { self.fd_cmask.as_ptr() };
                    if thz_session_set_fd_filters(s, r, c, nf) != THZ_OK { log::error!("flush: set_fd_filters failed"); return false; }
                }
                self.plugins_dirty = false;
            }
            if self.rois_dirty {
                let counts: Vec<usize> = self.rois.iter().map(|r| r.1.len()).collect();
                let flat: Vec<u64> = self.rois.iter().flat_map(|r| r.1.iter().flat_map(|v| [v.0 as u64, v.1 as u64])).collect();
                if thz_group_session_set_rois(self.session, self.rois.len(), counts.as_ptr(), flat.as_ptr()) != THZ_OK {
                    log::error!("flush: set_rois: {}", self.last_error());
                    return false;
                }
                self.rois_dirty = false;
                self.touch(POS_IFFT);
            }
            if self.dirty_from > POS_DECONVOLUTION { return true; } // nothing recorded since the last flush
            if thz_group_session_recompute(self.session, &self.pending, self.dirty_from as c_int, THZ_GATHER_SMALL) != THZ_OK {
                log::error!("flush: {}", self.last_error());
                return false;
            }
        }
        self.dirty_from = POS_DECONVOLUTION + 1;
        true
    }

    /// The Deconvolution stage over the WHOLE group (`thz_group_session_deconvolve`: on one GPU the session's own
    /// stage, on several the phased form: per-pixel parts on every member's rows, per-band iterations); everything in front of the stage is flushed first.  Progress
    /// and abort are forwarded live: the engine polls a plain `int` between iteration batches and writes its
    /// progress into a `float`; a watcher thread bridges them to `abort_flag` / `progress_lock`.
    pub fn deconvolve(&mut self, psf: &ThzPsf, cfg: &ThzDeconvCfg, progress_lock: &Arc<RwLock<Option<f32>>>,
                      abort_flag: &Arc<AtomicBool>) -> c_int {
        if !self.flush() { return THZ_ERR_NOT_READY; }
        let abort_i32 = Arc::new(AtomicI32::new(abort_flag.load(Ordering::Relaxed) as i32));
        let progress = Box::into_raw(Box::new(0f32));
        let stop = Arc::new(AtomicBool::new(false));
        let (a2, src, lock, stop2, p_addr) = (abort_i32.clone(), abort_flag.clone(), progress_lock.clone(), stop.clone(), progress as usize);
        let watcher = std::thread::spawn(move || {
            while !stop2.load(Ordering::Acquire) {
                if src.load(Ordering::Relaxed) { a2.store(1, Ordering::Relaxed); }
                let p = unsafe { std::ptr::read_volatile(p_addr as *const f32) };
                if let Ok(mut g) = lock.write() { *g = Some(p); }
                std::thread::sleep(std::time::Duration::from_millis(1));
            }
        });
        let rc = unsafe { thz_group_session_deconvolve(self.session, psf, cfg, abort_i32.as_ptr() as *const c_int, progress) };
        stop.store(true, Ordering::Release);
        let _ = watcher.join();
        unsafe { drop(Box::from_raw(progress)); }
        if let Ok(mut g) = progress_lock.write() { *g = None; }
        rc
    }

    // ---- results of the last flush
    /// image on the outputs' grid (`nx / s` x `ny / s`) -> (values, gx, gy)
    pub fn image(&self) -> Option<(Vec<f32>, usize, usize)> {
        if self.session.is_null() { return None; }
        unsafe {
            let (mut gx, mut gy) = (self.nx, self.ny);
            let (mut rows, mut cols) = (0usize, 0usize);
            for i in 0..thz_group_local_count(self.group) {
                let (mut r, mut c) = (0usize, 0usize);
                thz_session_grid(thz_group_session_member(self.session, i), &mut r, &mut c, ptr::null_mut(), ptr::null_mut());
                rows += r; cols = c;
            }
            if thz_group_local_count(self.group) == thz_group_world(self.group) { gx = rows; gy = cols; }
            let mut img = vec![0f32; gx * gy];
            let mut rc = thz_group_session_download(self.session, THZ_BUF_IMG, 0, gx * gy, img.as_mut_ptr() as *mut c_void);
            if rc == THZ_ERR_NOT_READY { // before the first recompute: the upload's image of the raw grid
                rc = thz_session_download(thz_group_session_member(self.session, 0), THZ_BUF_IMG, 0, gx * gy, img.as_mut_ptr() as *mut c_void);
            }
            if rc == THZ_OK { Some((img, gx, gy)) } else { None }
        }
    }

    /// pixel means of the ifft stage (`math_tools.rs:421-440`): (avg_fft, avg_signal_fft, avg_phase_fft)
    pub fn averages(&self) -> Option<(Vec<Complex32>, Vec<f32>, Vec<f32>)> {
        if self.session.is_null() { return None; }
        let nf = self.nt_out() / 2 + 1;
        let mut f = vec![Complex32::new(0.0, 0.0); nf];
        let (mut a, mut p) = (vec![0f32; nf], vec![0f32; nf]);
        let ok = unsafe {
            thz_group_session_download(self.session, THZ_BUF_AVG_FFT, 0, 1, f.as_mut_ptr() as *mut c_void) == THZ_OK
                && thz_group_session_download(self.session, THZ_BUF_AVG_AMPLITUDES, 0, 1, a.as_mut_ptr() as *mut c_void) == THZ_OK
                && thz_group_session_download(self.session, THZ_BUF_AVG_PHASES, 0, 1, p.as_mut_ptr() as *mut c_void) == THZ_OK
        };
        if ok { Some((f, a, p)) } else { None }
    }

    /// the member whose slab holds raw row `px` (`thz_host_slab`: the partition the library itself uses)
    fn owner_of(&self, px: usize) -> Option<(*mut ThzSession, usize)> {
        unsafe {
            let world = thz_group_world(self.group);
            for i in 0..thz_group_local_count(self.group) {
                let (mut x0, mut n) = (0usize, 0usize);
                thz_host_slab(self.nx, world, thz_group_rank(self.group, i), &mut x0, &mut n);
                if px >= x0 && px < x0 + n { return Some((thz_group_session_member(self.session, i), px - x0)); }
            }
        }
        None
    }

    /// `UpdateType::Plot` copy-out for pixel (px, py) of the RAW grid (`data_thread.rs:1337-1432`)
    pub fn plot(&self, px: usize, py: usize, out: &ThzPlotOut) -> bool {
        if self.session.is_null() { return false; }
        let (s, lx) = match self.owner_of(px) { Some(v) => v, None => return false };
        let sf = if self.pending.scale_factor > 1 { self.pending.scale_factor as usize } else { 1 };
        unsafe {
            if sf == 1 || thz_group_world(self.group) == 1 { return thz_session_plot(s, lx, py, out) == THZ_OK; }
            // Several slabs behind a scaling stage: the raw trace comes from the slab that holds row px, everything else
            // from the slab that holds the pixel's BLOCK — the one with the block's last raw row
            let mut raw = crate::math_tools_gpu::empty_plot_out();
            raw.signal = out.signal;
            if !out.signal.is_null() && thz_session_plot(s, lx, py, &raw) != THZ_OK { return false; }
            let mut rest = ThzPlotOut { ..*out };
            rest.signal = ptr::null_mut();
            match self.owner_of((px / sf) * sf + sf - 1) {
                Some((sb, lb)) => thz_session_plot(sb, lb, py, &rest) == THZ_OK,
                None => false,
            }
        }
    }

    /// one region's vectors (`math_tools.rs:473-543`, `data_thread.rs:1442-1482`)
    pub fn roi(&self, uuid: &str, out: &ThzRoiOut) -> bool {
        if self.session.is_null() { return false; }
        match self.rois.iter().position(|r| r.0 == uuid) {
            Some(i) => unsafe { thz_group_session_roi(self.session, i, out) == THZ_OK },
            None => false,
        }
    }

    pub fn nt_out(&self) -> usize {
        if self.session.is_null() { 0 } else { unsafe { thz_session_nt_out(thz_group_session_member(self.session, 0)) } }
    }
    pub fn time_out(&self) -> Vec<f32> {
        let mut t = vec![0f32; self.nt_out()];
        if !self.session.is_null() && !t.is_empty() { unsafe { thz_session_time_out(thz_group_session_member(self.session, 0), t.as_mut_ptr()); } }
        t
    }

    /// the 3-D tab's instances (`update_intensity_image`, `data_thread.rs:48-101`; `gui/threed_plot.rs:132-276`) of
    /// local member 0's slab — the whole cube on one GPU.  `InstanceData` and `ThzVoxelInstance` share their layout.
    pub fn voxels(&self, cfg: &ThzVoxelCfg, max_instances: u64, scaling: usize, orig: (usize, usize, usize))
                  -> Option<(Vec<ThzVoxelInstance>, f32, [f32; 3])> {
        if self.session.is_null() { return None; }
        unsafe {
            let s = thz_group_session_member(self.session, 0);
            let (mut n, mut thr, mut dims) = (0u64, 0f32, [0f32; 3]);
            if thz_session_voxels(s, cfg, max_instances, scaling as c_int, orig.0, orig.1, orig.2, ptr::null_mut(), 0, &mut n, &mut thr, dims.as_mut_ptr()) != THZ_OK {
                return None;
            }
            let mut out: Vec<ThzVoxelInstance> = Vec::with_capacity(n as usize);
            let cap = n;
            if thz_session_voxels(s, cfg, max_instances, scaling as c_int, orig.0, orig.1, orig.2, out.as_mut_ptr(), cap, &mut n, &mut thr, dims.as_mut_ptr()) != THZ_OK {
                return None;
            }
            out.set_len(n.min(cap) as usize);
            Some((out, thr, dims))
        }
    }
}

impl Drop for GpuEngine {
    fn drop(&mut self) {
        unsafe {
            if !self.session.is_null() { thz_group_session_destroy(self.session); }
            if !self.group.is_null() { thz_group_destroy(self.group); }
        }
    }
}
