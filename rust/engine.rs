//! Process-global GPU engine behind the data thread.
//!
//! `Filter` objects are cloned before every run and only `#[static_field]` members are copied back
//! (`data_thread.rs:1069-1078, 1322-1334`), so device state cannot live in a plugin struct: it lives here,
//! behind a `once_cell::Lazy` (`once_cell` is already a dependency, `Cargo.toml:69`).  One `thz_group` drives
//! every GPU of the node from the single data thread (`data_thread.rs:162-174`); with one GPU the group is
//! trivial and no RCCL is loaded.
//!
//! UNVERIFIED BY A COMPILER (no Rust toolchain in the authoring image); mirrors `include/thzgpu.h` via `ffi.rs`.
use super::ffi::*;
use ndarray::{Array1, Array2};
use num_complex::Complex32;
use once_cell::sync::Lazy;
use std::ffi::CStr;
use std::os::raw::{c_int, c_void};
use std::ptr;
use std::sync::atomic::{AtomicBool, AtomicI32, Ordering};
use std::sync::{Arc, Mutex, RwLock};

/// `THZGPU_DEVICES=0,1,2,3` selects the GPUs; default: device 0.
fn devices_from_env() -> Vec<c_int> {
    std::env::var("THZGPU_DEVICES")
        .ok()
        .map(|s| s.split(',').filter_map(|t| t.trim().parse().ok()).collect::<Vec<c_int>>())
        .filter(|v| !v.is_empty())
        .unwrap_or_else(|| vec![0])
}

pub static ENGINE: Lazy<Mutex<GpuEngine>> = Lazy::new(|| Mutex::new(GpuEngine::new(&devices_from_env())));

#[derive(Debug)]
pub struct GpuError(pub c_int, pub String);

pub struct GpuEngine {
    group: *mut ThzGroup,
    session: *mut ThzGroupSession,
    /// what the stage calls of the current walk have recorded (see `record_*`)
    pub pending: ThzChainCfg,
    /// lowest chain position touched since the last flush: the `start_idx` of `UpdateType::Filter`
    dirty_from: usize,
    pub nx: usize,
    pub ny: usize,
    pub nt: usize,
    fd_real: Option<Vec<f32>>,
    fd_cmask: Option<Vec<f32>>,
    /// `abort_flag` as the engine polls it (an `AtomicBool` has no C-visible layout guarantee)
    abort_i32: Arc<AtomicI32>,
}

// the raw handles are only touched under the Mutex around the engine
unsafe impl Send for GpuEngine {}

impl GpuEngine {
    pub fn new(devices: &[c_int]) -> Self {
        let mut group = ptr::null_mut();
        let rc = unsafe { thz_group_create(devices.as_ptr(), devices.len() as c_int, &mut group) };
        if rc != THZ_OK {
            // no CPU compute path exists in the engine: the caller keeps the reference's rayon path
            log::error!("thz_group_create({devices:?}) failed with {rc}: GPU path disabled");
            group = ptr::null_mut();
        }
        let mut pending: ThzChainCfg = unsafe { std::mem::zeroed() };
        pending.scale_factor = 1;
        pending.want_means = 1;
        GpuEngine { group, session: ptr::null_mut(), pending, dirty_from: 1, nx: 0, ny: 0, nt: 0, fd_real: None,
                    fd_cmask: None, abort_i32: Arc::new(AtomicI32::new(0)) }
    }

    pub fn available(&self) -> bool { !self.group.is_null() }

    fn err(&self, rc: c_int) -> GpuError {
        let msg = unsafe { CStr::from_ptr(thz_group_last_error(self.group)) }.to_string_lossy().into_owned();
        GpuError(rc, msg)
    }
    fn check(&self, rc: c_int) -> Result<(), GpuError> { if rc < 0 { Err(self.err(rc)) } else { Ok(()) } }

    /// `ConfigCommand::OpenFile` (`data_thread.rs:176-…`, `io.rs:576-628`): the cube goes to the device(s) once;
    /// the per-trace bias subtraction and the first intensity image happen there.
    pub fn open_scan(&mut self, cube: &ndarray::Array3<f32>, time: &Array1<f32>, dx: f32, dy: f32) -> Result<Array2<f32>, GpuError> {
        let (nx, ny, nt) = cube.dim();
        unsafe {
            if !self.session.is_null() { thz_group_session_destroy(self.session); self.session = ptr::null_mut(); }
            let t = time.as_slice().expect("contiguous time axis");
            self.check(thz_group_session_create(self.group, nx, ny, nt, t.as_ptr(), dx, dy, &mut self.session))?;
            let c = cube.as_slice().expect("C-order cube (data_container.rs:136-151)");
            self.check(thz_group_session_upload(self.session, c.as_ptr(), 1))?;
            self.check(thz_chain_cfg_default(t.as_ptr(), nt, &mut self.pending))?;
        }
        self.nx = nx; self.ny = ny; self.nt = nt; self.dirty_from = 1;
        self.fd_real = None; self.fd_cmask = None;
        self.image()
    }

    // ---- stage calls record their parameters; chain positions as in main.rs:182-247 ("initial" = 0)
    pub fn touch(&mut self, chain_position: usize) { self.dirty_from = self.dirty_from.min(chain_position.max(1)); }
    pub fn record_scaling(&mut self, scale_factor: usize) { self.pending.scale_factor = scale_factor as i32; self.touch(1); }
    pub fn record_tilt(&mut self, active: bool, tilt_x: f64, tilt_y: f64) {
        self.pending.tilt_active = active as i32; self.pending.tilt_x_deg = tilt_x; self.pending.tilt_y_deg = tilt_y; self.touch(2);
    }
    pub fn record_td_before(&mut self, active: bool, low: f64, high: f64, width: f64) {
        self.pending.td_before_active = active as i32; self.pending.td_before_low = low; self.pending.td_before_high = high;
        self.pending.td_before_width = width; self.touch(3);
    }
    pub fn record_fft(&mut self, window_type: i32, lower: f32, upper: f32) {
        self.pending.fft_window = ThzWindowCfg { type_: window_type, lower, upper }; self.touch(4);
    }
    pub fn record_fd(&mut self, active: bool, low: f64, high: f64, width: f64) {
        self.pending.fd_active = active as i32; self.pending.fd_low = low; self.pending.fd_high = high;
        self.pending.fd_width = width; self.touch(5);
    }
    /// further Frequency-domain plugins: K14 (real) and K13 (complex, interleaved) per-bin multipliers
    /// (each plugin sets its own kind; the walk starts by clearing both: `begin_walk`)
    pub fn record_fd_plugins(&mut self, real_mask: Option<Vec<f32>>, cmask: Option<Vec<f32>>) {
        if real_mask.is_some() { self.fd_real = real_mask; }
        if cmask.is_some() { self.fd_cmask = cmask; }
        self.touch(5);
    }
    /// top of a stage walk that starts at or in front of the Frequency plugins: inactive ones must not linger
    pub fn begin_walk(&mut self, start_idx: usize) {
        if start_idx <= 5 { self.fd_real = None; self.fd_cmask = None; }
        self.touch(start_idx);
    }
    pub fn record_ifft(&mut self) { self.touch(6); }
    pub fn record_td_after(&mut self, active: bool, low: f64, high: f64, width: f64) {
        self.pending.td_after_active = active as i32; self.pending.td_after_low = low; self.pending.td_after_high = high;
        self.pending.td_after_width = width; self.touch(7);
    }

    /// After the stage loop (`data_thread.rs:1229`): one recompute from the lowest touched position.
    pub fn flush(&mut self) -> Result<(), GpuError> {
        if self.session.is_null() { return Err(GpuError(THZ_ERR_NOT_READY, "no file open".into())); }
        let nf = self.nt / 2 + 1;
        unsafe {
            for i in 0..thz_group_local_count(self.group) {
                let s = thz_group_session_member(self.session, i);
                let r = self.fd_real.as_ref().map_or(ptr::null(), |v| v.as_ptr());
                let c = self.fd_cmask.as_ref().map_or(ptr::null(), |v| v.as_ptr());
                self.check(thz_session_set_fd_filters(s, r, c, if r.is_null() && c.is_null() { 0 } else { nf }))?;
            }
            self.check(thz_group_session_recompute(self.session, &self.pending, self.dirty_from as c_int, THZ_GATHER_SMALL))?;
        }
        self.dirty_from = 9;
        Ok(())
    }

    /// `img_lock` content (`data_thread.rs:1310-1315`)
    pub fn image(&self) -> Result<Array2<f32>, GpuError> {
        let mut img = Array2::<f32>::zeros((self.nx, self.ny));
        let rc = unsafe { thz_group_session_download(self.session, THZ_BUF_IMG, 0, self.nx * self.ny,
                                                     img.as_mut_ptr() as *mut c_void) };
        if rc == THZ_ERR_NOT_READY {      // before the first recompute: the upload's image of the raw grid
            let s = unsafe { thz_group_session_member(self.session, 0) };
            self.check(unsafe { thz_session_download(s, THZ_BUF_IMG, 0, self.nx * self.ny, img.as_mut_ptr() as *mut c_void) })?;
            return Ok(img);
        }
        self.check(rc)?;
        Ok(img)
    }

    /// pixel means of the ifft stage (`math_tools.rs:421-440`): (avg_fft, avg_signal_fft, avg_phase_fft)
    pub fn averages(&self) -> Result<(Array1<Complex32>, Array1<f32>, Array1<f32>), GpuError> {
        let nf = self.nt / 2 + 1;
        let mut f = vec![Complex32::new(0.0, 0.0); nf];
        let (mut a, mut p) = (vec![0f32; nf], vec![0f32; nf]);
        unsafe {
            self.check(thz_group_session_download(self.session, THZ_BUF_AVG_FFT, 0, 1, f.as_mut_ptr() as *mut c_void))?;
            self.check(thz_group_session_download(self.session, THZ_BUF_AVG_AMPLITUDES, 0, 1, a.as_mut_ptr() as *mut c_void))?;
            self.check(thz_group_session_download(self.session, THZ_BUF_AVG_PHASES, 0, 1, p.as_mut_ptr() as *mut c_void))?;
        }
        Ok((Array1::from(f), Array1::from(a), Array1::from(p)))
    }

    /// `UpdateType::Plot` copy-out for the selected pixel (`data_thread.rs:1337-1432`): the pixel's slab owner
    /// serves it (`thz_host_slab` tells which member holds row `px`)
    pub fn plot(&self, px: usize, py: usize, out: &ThzPlotOut) -> Result<(), GpuError> {
        let world = unsafe { thz_group_world(self.group) };
        for i in 0..unsafe { thz_group_local_count(self.group) } {
            let (mut x0, mut n) = (0usize, 0usize);
            unsafe { thz_host_slab(self.nx, world, thz_group_rank(self.group, i), &mut x0, &mut n) };
            if px >= x0 && px < x0 + n {
                let s = unsafe { thz_group_session_member(self.session, i) };
                return self.check(unsafe { thz_session_plot(s, px - x0, py, out) });
            }
        }
        Err(GpuError(THZ_ERR_INVALID, "pixel outside every slab".into()))
    }

    /// Deconvolution stage (single-GPU sessions; `FilterDomain::TimeAfterFFTPrioLast`): progress and abort are
    /// forwarded live — the engine polls `abort` between iteration batches and writes `progress` as it goes.
    pub fn deconvolve(&mut self, psf: &ThzPsf, cfg: &ThzDeconvCfg, progress_lock: &Arc<RwLock<Option<f32>>>,
                      abort_flag: &Arc<AtomicBool>) -> Result<c_int, GpuError> {
        let s = unsafe { thz_group_session_member(self.session, 0) };
        self.abort_i32.store(abort_flag.load(Ordering::Relaxed) as i32, Ordering::Relaxed);
        let progress = Box::new(0f32);
        let progress_ptr = Box::into_raw(progress);
        // a watcher copies the AtomicBool into the i32 the engine polls and the engine's f32 into the lock
        let (abort_i32, abort_src, lock, stop) = (self.abort_i32.clone(), abort_flag.clone(), progress_lock.clone(), Arc::new(AtomicBool::new(false)));
        let stop2 = stop.clone();
        let p_addr = progress_ptr as usize;
        let watcher = std::thread::spawn(move || {
            while !stop2.load(Ordering::Relaxed) {
                abort_i32.store(abort_src.load(Ordering::Relaxed) as i32, Ordering::Relaxed);
                let p = unsafe { std::ptr::read_volatile(p_addr as *const f32) };
                if let Ok(mut g) = lock.write() { *g = Some(p); }
                std::thread::sleep(std::time::Duration::from_millis(20));
            }
        });
        let rc = unsafe { thz_session_deconvolve(s, psf, cfg, self.abort_i32.as_ptr() as *const c_int, progress_ptr) };
        stop.store(true, Ordering::Relaxed);
        let _ = watcher.join();
        unsafe { drop(Box::from_raw(progress_ptr)); }
        if let Ok(mut g) = progress_lock.write() { *g = None; }
        if rc < 0 { return Err(self.err(rc)); }
        Ok(rc)
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
