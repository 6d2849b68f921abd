"""thz_session (resident cube + whole-chain recompute, include/thzgpu.h) against the
oracle chain: what the data thread does for OpenFile + UpdateType::Filter(idx)
(data_thread.rs:1023-1334), with defaults, changed sliders and a tilted scan."""
import numpy as np
import pytest

import oracle_binding as ob
import synth
import thz_image_explorer_amd as pkg
from test_gpu_parity import TOL, phase_ok, rel

pytestmark = pytest.mark.gpu


def oracle_chain(cube, time, cfg, dx=1.0, dy=1.0):
    """stage-by-stage oracle with the parameters of a thz_chain_cfg"""
    d, t = cube, time
    if cfg.tilt_active:
        _, t, d = ob.tilt(d, t, cfg.tilt_x_deg, cfg.tilt_y_deg, dx, dy)
    if cfg.td_before_active:
        d, _, _ = ob.td_bandpass(d, t, cfg.td_before_low, cfg.td_before_high, cfg.td_before_width)
    st = ob.fft_stage(d, t, cfg.fft_window.type, cfg.fft_window.lower, cfg.fft_window.upper)
    fft, amp = st["fft"], st["amplitudes"]
    freq = ob.frequency_axis(t)
    if cfg.fd_active:
        fft, amp = ob.fd_bandpass(fft, amp, freq, cfg.fd_low, cfg.fd_high, cfg.fd_width)
    avg = dict(fft=ob.pixel_mean(fft, 2), amp=ob.pixel_mean(amp), ph=ob.pixel_mean(st["phases"]))
    out, nerr = ob.ifft_stage(fft, t.size)
    assert nerr == 0
    if cfg.td_after_active:
        out, _, _ = ob.td_bandpass(out, t, cfg.td_after_low, cfg.td_after_high, cfg.td_after_width)
    return dict(time=t, fft=fft, amp=amp, ph=st["phases"], amp_unmasked=st["amplitudes"], data=out,
                img=ob.intensity(out), avg=avg)


def check(sess, ref, nx, ny):
    nto = sess.nt_out
    assert nto == ref["time"].size
    assert np.array_equal(sess.time_out(), ref["time"])
    nf = nto // 2 + 1
    scale = np.abs(ref["fft"]).max()
    assert rel(sess.download(pkg.BUF_FFT).reshape(nx, ny, nf, 2), ref["fft"], scale) < TOL
    assert rel(sess.download(pkg.BUF_AMPLITUDES).reshape(nx, ny, nf), ref["amp"], scale) < TOL
    assert rel(sess.download(pkg.BUF_DATA).reshape(nx, ny, nto), ref["data"]) < TOL
    assert rel(sess.download(pkg.BUF_IMG).reshape(nx, ny), ref["img"]) < TOL
    assert rel(sess.download(pkg.BUF_AVG_FFT), ref["avg"]["fft"], np.abs(ref["avg"]["fft"]).max()) < TOL
    assert rel(sess.download(pkg.BUF_AVG_AMPLITUDES), ref["avg"]["amp"]) < TOL


@pytest.mark.parametrize("shape", [(6, 8, 1024), (3, 4, 256), (4, 4, 1001)])
def test_session_default_chain_and_slider_changes(engine, shape):
    nx, ny, nt = shape
    time, _ = synth.make_cube(1, 1, nt)
    ids = np.arange(nx * ny, dtype=np.uint64)
    raw = synth.make_traces(ids, nt, subtract_bias=False).reshape(nx, ny, nt)
    sess = pkg.Session(engine, nx, ny, time)
    try:
        sess.upload(raw, subtract_bias=True)
        cube = ob.subtract_bias(raw)
        assert np.array_equal(sess.download(pkg.BUF_RAW).reshape(nx, ny, nt), cube)
        assert rel(sess.download(pkg.BUF_IMG).reshape(nx, ny), ob.intensity(cube)) < TOL
        cfg = pkg.chain_cfg_default(time)
        assert cfg.td_before_low == float(time[0]) and cfg.td_after_high == float(time[-1])
        sess.recompute(cfg)
        check(sess, oracle_chain(cube, time, cfg), nx, ny)
        # selected-pixel download == slice of the full download
        full = sess.download(pkg.BUF_DATA)
        assert np.array_equal(sess.download(pkg.BUF_DATA, 5, 2), full[5:7])
        # slider changes: UpdateType::Filter(idx) with new parameters
        cfg.fd_low, cfg.fd_high = 0.4, 2.5
        cfg.td_before_low = float(time[0]) + 3.0
        cfg.td_after_high = float(time[-1]) - 4.0
        cfg.fft_window.type = 3  # Hamming
        sess.recompute(cfg)
        check(sess, oracle_chain(cube, time, cfg), nx, ny)
        # filters switched off (Filter::config().active == false, data_thread.rs:1140-1153)
        cfg.fd_active = 0
        cfg.td_before_active = 0
        cfg.tilt_active = 0
        sess.recompute(cfg)
        check(sess, oracle_chain(cube, time, cfg), nx, ny)
    finally:
        sess.close()


def test_session_tilted_scan_changes_trace_length(engine):
    nx, ny, nt = 6, 5, 1024
    time, cube = synth.make_cube(nx, ny, nt)
    sess = pkg.Session(engine, nx, ny, time, dx=0.5, dy=0.5)
    try:
        sess.upload(cube, subtract_bias=False)
        cfg = pkg.chain_cfg_default(time)
        cfg.tilt_x_deg, cfg.tilt_y_deg = 2.0, -1.0
        sess.recompute(cfg)
        ref = oracle_chain(cube, time, cfg, 0.5, 0.5)
        assert sess.nt_out > nt
        check(sess, ref, nx, ny)
        # back to zero tilt: outputs return to the original length
        cfg.tilt_x_deg = cfg.tilt_y_deg = 0.0
        sess.recompute(cfg)
        assert sess.nt_out == nt
        check(sess, oracle_chain(cube, time, cfg, 0.5, 0.5), nx, ny)
    finally:
        sess.close()


def test_session_plot_copy_out(engine):
    """UpdateType::Plot, data_thread.rs:1337-1436: selected-pixel and average vectors in one call"""
    nx, ny, nt = 5, 6, 1024
    time, cube = synth.make_cube(nx, ny, nt)
    for tilt in (0.0, 1.5):
        sess = pkg.Session(engine, nx, ny, time, dx=0.5, dy=0.5)
        try:
            sess.upload(cube, subtract_bias=False)
            with pytest.raises(pkg.ThzError):
                sess.plot(1, 1)                                # nothing computed yet
            assert np.array_equal(sess.plot(1, 1, want=["signal"])["signal"], cube[1, 1])
            cfg = pkg.chain_cfg_default(time)
            cfg.tilt_x_deg = tilt
            sess.recompute(cfg)
            ref = oracle_chain(cube, time, cfg, 0.5, 0.5)
            px, py = 3, 4
            got = sess.plot(px, py)
            scale = np.abs(ref["fft"]).max()
            assert np.array_equal(got["signal"], cube[px, py])
            assert rel(got["signal_fft"], ref["amp_unmasked"][px, py], scale) < TOL
            assert phase_ok(got["phase_fft"][None], ref["ph"][px, py][None], ref["amp_unmasked"][px, py][None])
            assert rel(got["filtered_signal"], ref["data"][px, py], np.abs(ref["data"]).max()) < TOL
            assert rel(got["filtered_signal_fft"], ref["amp"][px, py], scale) < TOL
            assert phase_ok(got["filtered_phase_fft"][None], ref["ph"][px, py][None], ref["amp_unmasked"][px, py][None])
            assert rel(got["avg_signal"], ob.pixel_mean(ref["data"]), np.abs(ref["data"]).max()) < TOL
            assert rel(got["avg_signal_fft"], ref["avg"]["amp"]) < TOL
            with pytest.raises(pkg.ThzError):
                sess.plot(nx, 0)
        finally:
            sess.close()


def test_session_tilted_4096_scan(engine):
    """a tilted 4096-sample scan leaves the power-of-two lengths: nt_out = 4096 + 2 * steps goes
    through the eight-run chirp-z kernels"""
    nx, ny, nt = 2, 3, 4096
    time, cube = synth.make_cube(nx, ny, nt)
    sess = pkg.Session(engine, nx, ny, time, dx=2.0, dy=2.0)
    try:
        sess.upload(cube, subtract_bias=False)
        cfg = pkg.chain_cfg_default(time)
        cfg.tilt_x_deg, cfg.tilt_y_deg = 3.0, 1.0
        sess.recompute(cfg)
        assert 4096 < sess.nt_out < 8192
        assert engine.kernel_variant().startswith("fb8-")
        check(sess, oracle_chain(cube, time, cfg, 2.0, 2.0), nx, ny)
    finally:
        sess.close()


def test_session_voxels(engine):
    """update_intensity_image's 3-D part on the session's final cube (data_thread.rs:82-101)"""
    nx, ny, nt = 10, 9, 1024
    time, cube = synth.make_cube(nx, ny, nt)
    cube = (cube * np.float32(4.0)).astype(np.float32)     # envelope maxima above the opacity threshold
    sess = pkg.Session(engine, nx, ny, time)
    try:
        sess.upload(cube, subtract_bias=False)
        vcfg = pkg.voxel_cfg_default()
        with pytest.raises(pkg.ThzError):
            sess.voxels(vcfg)                               # nothing computed yet
        cfg = pkg.chain_cfg_default(time)
        sess.recompute(cfg)
        inst, thr, dims = sess.voxels(vcfg, max_instances=3000)
        final = sess.download(pkg.BUF_DATA).reshape(nx, ny, nt)
        op_gpu = sess.download(pkg.BUF_OPACITY).reshape(nx, ny, nt)
        assert np.abs(op_gpu - ob.voxel_opacity(final)).max() < 1e-5
        assert thr == ob.voxel_threshold(op_gpu, 3000)
        ref, rdims = ob.voxel_instances(op_gpu, thr, float(time[-1] - time[0]), 1, (nx, ny, nt))
        assert dims == rdims and len(inst) == len(ref) >= 3000
        assert np.array_equal(inst["position"], ref["position"])
        assert np.array_equal(inst["color"][:, 3], ref["color"][:, 3])
        # a buffer smaller than the count: complete count is still reported by the first (sizing) call
        few, _, _ = sess.voxels(vcfg, max_instances=3000, capacity=10)
        assert len(few) == 10 and np.array_equal(few["position"], ref["position"][:10])
    finally:
        sess.close()


def test_real_thz_file_through_session(engine):
    """a real sample file of the reference (single-pulse .thz, nt = 1001), opened by the dotTHz reader,
    through OpenFile's preprocessing and the default chain — against the oracle on the same trace"""
    import os
    from thz_image_explorer_amd import io_binding as tio
    if not tio.available():
        pytest.skip("libthzio.so not built")
    path = os.path.join(os.path.dirname(__file__), "golden", "knife_edge_2groups.thz")
    with tio.ScanFile(path) as f:
        time, raw, g = f.time(), f.cube(), f.geometry()
    assert raw.shape == (1, 1, 1001) and g.has_dx
    sess = pkg.Session(engine, 1, 1, time, g.dx, g.dy)
    try:
        sess.upload(raw, subtract_bias=True)
        cfg = pkg.chain_cfg_default(time)
        sess.recompute(cfg)
        check(sess, oracle_chain(ob.subtract_bias(raw), time, cfg, g.dx, g.dy), 1, 1)
    finally:
        sess.close()


def test_session_scaling_stage(engine):
    """ConfigContainer.scale_factor: the chain's first stage (math_tools.rs:242-310) — block means of the raw
    cube incl. the ragged-edge rule, dx / dy grown by s, everything behind it on the block grid"""
    nx, ny, nt, sf = 9, 7, 1024, 2
    time, cube = synth.make_cube(nx, ny, nt)
    sess = pkg.Session(engine, nx, ny, time, 0.5, 0.25)
    try:
        sess.upload(cube, subtract_bias=False)
        assert sess.grid() == (nx, ny, 0.5, 0.25)
        cfg = pkg.chain_cfg_default(time)
        assert cfg.scale_factor == 1
        cfg.scale_factor = sf
        cfg.tilt_x_deg, cfg.tilt_y_deg = 1.5, -0.5            # uses the grown dx / dy
        sess.recompute(cfg)
        gx, gy, gdx, gdy = sess.grid()
        assert (gx, gy, gdx, gdy) == (nx // sf, ny // sf, 1.0, 0.5)
        small = ob.scale3d(cube, sf)
        ref = oracle_chain(small, time, cfg, gdx, gdy)
        assert ref["time"].size > nt                            # the tilt really extended the axis
        check(sess, ref, gx, gy)
        # plot copy-out: raw trace of the pixel itself, processed traces of its block
        po = sess.plot(5, 3)
        assert np.array_equal(po["signal"], cube[5, 3])
        assert np.array_equal(po["filtered_signal"], sess.download(pkg.BUF_DATA).reshape(gx, gy, -1)[5 // sf, 3 // sf])
        with pytest.raises(pkg.ThzError):
            sess.plot(8, 0)                                     # ragged edge: no block behind this pixel
        # a factor larger than a side leaves the grid alone (:251-256); so does going back to 1
        for f in (8, 1):
            cfg.scale_factor = f
            sess.recompute(cfg)
            assert sess.grid() == (nx, ny, 0.5, 0.25)
        check(sess, oracle_chain(cube, time, cfg, 0.5, 0.25), nx, ny)
    finally:
        sess.close()


def test_session_deconvolution_stage(engine):
    """the chain's last stage and its gating (data_thread.rs:1080, 1139-1149, 1186-1188): updating the
    Deconvolution filter deconvolves the Time Band Pass output; updating any other filter passes it through"""
    import os
    from test_gpu_deconv import _bar_target_cube
    nx, ny, nt = 32, 32, 256
    time, cube = _bar_target_cube(nx, ny, nt)
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "psf_sample.npz"))
    psf, opsf = pkg.psf_from_npz(z), ob.psf_from_npz(z)
    dcfg = pkg.DeconvCfg(20, 6, 0.4, 3.0, 0.5)
    sess = pkg.Session(engine, nx, ny, time, 0.5, 0.5)
    try:
        sess.upload(cube, subtract_bias=False)
        with pytest.raises(pkg.ThzError):
            sess.deconvolve(psf, dcfg)                      # nothing computed yet
        cfg = pkg.chain_cfg_default(time)
        sess.recompute(cfg)
        stage_in = sess.download(pkg.BUF_DATA).reshape(nx, ny, nt)
        img_in = sess.download(pkg.BUF_IMG).reshape(nx, ny)
        assert sess.deconvolve(psf, dcfg) == 0
        out = sess.download(pkg.BUF_DATA).reshape(nx, ny, nt)
        img = sess.download(pkg.BUF_IMG).reshape(nx, ny)
        rc, oref, oimg, _, _ = ob.deconvolution(stage_in, time, 0.5, 0.5, opsf, 20, 6, 0.4, 3.0, 0.5)
        assert rc == 0
        assert np.abs(out - oref).max() / np.abs(oref).max() < 2e-4
        assert np.abs(img - oimg).max() / oimg.max() < 5e-4
        assert np.abs(out - stage_in).max() / np.abs(stage_in).max() > 1e-2
        # the plot copy-out and a second update read the same stage input, not the previous result
        po = sess.plot(3, 4)
        assert np.array_equal(po["filtered_signal"], out[3, 4])
        assert sess.deconvolve(psf, dcfg) == 0
        assert np.array_equal(sess.download(pkg.BUF_DATA).reshape(nx, ny, nt), out)
        # a guard (fewer than 2 bands) hands the input through
        assert sess.deconvolve(psf, pkg.DeconvCfg(20, 1, 0.4, 3.0, 0.5)) == 1
        assert np.array_equal(sess.download(pkg.BUF_DATA).reshape(nx, ny, nt), stage_in)
        assert sess.deconvolve(psf, dcfg) == 0
        # another filter is updated: the deconvolution is not re-run, its stage passes the input through
        sess.recompute(cfg)
        assert np.array_equal(sess.download(pkg.BUF_DATA).reshape(nx, ny, nt), stage_in)
        assert np.array_equal(sess.download(pkg.BUF_IMG).reshape(nx, ny), img_in)
    finally:
        sess.close()


def test_session_download_bounds_and_missing_means(engine):
    time, cube = synth.make_cube(2, 2, 256)
    sess = pkg.Session(engine, 2, 2, time)
    try:
        sess.upload(cube, subtract_bias=False)
        cfg = pkg.chain_cfg_default(time)
        cfg.want_means = 0
        sess.recompute(cfg)
        with pytest.raises(pkg.ThzError):
            sess.download(pkg.BUF_AVG_FFT)
        with pytest.raises(pkg.ThzError):
            sess.download(pkg.BUF_DATA, 3, 2)
    finally:
        sess.close()


def test_session_outputs_absent_until_recomputed(engine):
    """after a scaled recompute the output buffers are sized for the block grid; an upload resets
    the session's grid to the raw one — the outputs must then read as absent (THZ_ERR_NOT_READY), not be copied
    past their end; a pixel range beyond the recompute's grid is THZ_ERR_INVALID"""
    nx, ny, nt = 8, 6, 256
    time, cube = synth.make_cube(nx, ny, nt)
    sess = pkg.Session(engine, nx, ny, time)
    try:
        outputs = (pkg.BUF_FFT, pkg.BUF_AMPLITUDES, pkg.BUF_PHASES, pkg.BUF_DATA, pkg.BUF_AVG_FFT)
        sess.upload(cube, subtract_bias=False)
        for which in outputs:   # nothing computed yet
            with pytest.raises(pkg.ThzError) as e:
                sess.download(which)
            assert e.value.code == -4
        assert sess.download(pkg.BUF_IMG).size == nx * ny    # the upload's image of the raw grid
        cfg = pkg.chain_cfg_default(time)
        cfg.scale_factor = 2
        sess.recompute(cfg)
        gx, gy, _, _ = sess.grid()
        assert (gx, gy) == (nx // 2, ny // 2)
        assert sess.download(pkg.BUF_DATA).size == gx * gy * nt
        with pytest.raises(pkg.ThzError) as e:   # the raw grid's pixel count no longer fits
            sess.download(pkg.BUF_DATA, 0, nx * ny)
        assert e.value.code == -1
        sess.upload(cube, subtract_bias=False)   # grid is the raw one again, outputs are void
        for which in outputs:
            with pytest.raises(pkg.ThzError) as e:
                sess.download(which, 0, nx * ny) if which != pkg.BUF_AVG_FFT else sess.download(which)
            assert e.value.code == -4
    finally:
        sess.close()


@pytest.mark.parametrize("shape", [(6, 8, 1024), (4, 5, 1001)])
def test_session_recompute_from_each_chain_position(engine, shape):
    """UpdateType::Filter(start_idx) (data_thread.rs:1090-1105): one slider changed per chain position, the walk
    started at that position (what UpdateFilter(uuid) / the fft-window commands send, :813-836, 907-921), every
    time against the oracle's stage-by-stage chain with the new settings.  Positions 1-5 run the whole chain, 6
    and 7 only the tail on the resident spectrum — with identical results to a recompute from the front."""
    nx, ny, nt = shape
    time, cube = synth.make_cube(nx, ny, nt)
    sess = pkg.Session(engine, nx, ny, time)
    try:
        sess.upload(cube, subtract_bias=False)
        cfg = pkg.chain_cfg_default(time)
        sess.recompute(cfg)
        t0, t1 = float(time[0]), float(time[-1])
        edits = [(3, lambda c: setattr(c, "td_before_low", t0 + 4.0)),       # Time Band Pass
                 (3, lambda c: setattr(c.fft_window, "upper", 5.0)),          # fft window commands send fft_index = 3
                 (5, lambda c: setattr(c, "fd_high", 3.5)),                   # Frequency Band Pass
                 (7, lambda c: setattr(c, "td_after_high", t1 - 6.0)),        # Time Band Pass (after)
                 (6, lambda c: None),                                         # ifft re-run, nothing changed
                 (7, lambda c: setattr(c, "td_after_width", 1.5)),
                 (2, lambda c: setattr(c, "tilt_active", 0)),                 # Tilt Compensation switched off
                 (7, lambda c: setattr(c, "td_after_low", t0 + 2.0))]
        for pos, edit in edits:
            edit(cfg)
            sess.recompute(cfg, pos)
            check(sess, oracle_chain(cube, time, cfg), nx, ny)
            tail = sess.download(pkg.BUF_DATA).copy()
            img = sess.download(pkg.BUF_IMG).copy()
            if pos >= 6:     # the tail-only path writes what the full chain writes (2e-6: fused vs stand-alone inverse)
                sess.recompute(cfg, 1)
                full = sess.download(pkg.BUF_DATA)
                assert np.abs(tail - full).max() <= 2e-6 * np.abs(full).max()
                assert np.abs(img - sess.download(pkg.BUF_IMG)).max() <= 2e-6 * img.max()
        # a tail start whose front settings differ from the last full recompute falls back to the full chain
        cfg.fd_low = 0.5
        sess.recompute(cfg, 7)
        check(sess, oracle_chain(cube, time, cfg), nx, ny)
    finally:
        sess.close()


@pytest.mark.parametrize("shape", [(4, 4, 1001), (3, 5, 300), (3, 3, 512)])
def test_session_and_pipeline_ex_on_the_generic_kernels(engine, shape):
    """thz_set_kernel_family(1): every length on the G kernels.  A length that is not a power of two is then a
    generic chirp-z plan, for which the fused entry point has no single kernel: thz_pipeline_ex (what every session
    recompute goes through) must run it as forward + inverse launches, as thz_pipeline does — it used to fall
    through to the power-of-two chain and read out of bounds."""
    nx, ny, nt = shape
    time, cube = synth.make_cube(nx, ny, nt)
    npix, nf = nx * ny, nt // 2 + 1
    chain_p, chain = synth.default_chain(time), synth.oracle_chain(time)
    ref = ob.run_pipeline(cube, time, chain)
    scale = np.abs(ref["fft"]).max()
    e = engine
    e.set_kernel_family(1)
    try:
        e.set_time_axis(time)
        assert e.kernel_variant().startswith("g-")
        bufs = [e.to_device(cube), e.to_device(chain_p["w_pre"]), e.to_device(chain_p["fd_mask"]), e.to_device(chain_p["w_post"]),
                e.empty((npix, nf, 2)), e.empty((npix, nf)), e.empty((npix, nf)), e.empty((npix, nt)), e.empty((npix,)),
                e.empty((2 * nf,))]
        d_raw, d_pre, d_fd, d_post, d_fft, d_amp, d_ph, d_out, d_img, d_sums = bufs
        e.pipeline_ex(npix, d_raw, d_pre, d_fd, None, d_post, d_fft, d_amp, d_ph, d_out, d_img, d_sums)
        amp = d_amp.download((npix, nf), np.float32)
        assert rel(d_fft.download((npix, nf, 2), np.float32), ref["fft"].reshape(npix, nf, 2), scale) < TOL
        assert rel(amp, ref["amplitudes"].reshape(npix, nf), scale) < TOL
        assert rel(d_out.download((npix, nt), np.float32), ref["data"].reshape(npix, nt)) < TOL
        assert rel(d_img.download((npix,), np.float32), ref["img"].ravel()) < TOL
        sa = amp.astype(np.float64).sum(0)
        assert np.abs(d_sums.download((2 * nf,), np.float32)[:nf] - sa).max() <= 2e-6 * np.abs(sa).max()
        for b in bufs:
            b.free()
        sess = pkg.Session(e, nx, ny, time)
        try:
            sess.upload(cube, subtract_bias=False)
            cfg = pkg.chain_cfg_default(time)
            sess.recompute(cfg)
            check(sess, oracle_chain(cube, time, cfg), nx, ny)
        finally:
            sess.close()
    finally:
        e.set_kernel_family(0)
