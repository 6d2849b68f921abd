"""thz_group / thz_group_session (include/thzgpu.h, "Multi-GPU"): x-slab tiles inside the library.

A one-GPU box cannot hold two RCCL ranks (RCCL refuses two ranks on one device), so these tests run the
group in its same-device form — n members on device 0, collectives as device-local copies — which exercises
everything except the fabric: the slab rule, per-slab sessions, C2 (sum of the slabs' pixel sums), C1 (gather
in rank order), the three gather levels.  The RCCL calls themselves are covered by the driver's multi-GPU
bench (bench.py --gpus N goes through the same entry points)."""
import numpy as np
import pytest

import oracle_binding as ob
import synth
import thz_image_explorer_amd as pkg
from test_gpu_parity import TOL, rel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("members", [1, 2, 3])
def test_group_collectives_same_device(members):
    with pkg.Group(devices=[0] * members) as g:
        assert g.world == members and g.ranks == list(range(members))
        engs = [g.engine(i) for i in range(members)]
        rng = np.random.default_rng(members)
        vals = [rng.standard_normal(1000).astype(np.float32) for _ in range(members)]
        bufs = [e.to_device(v) for e, v in zip(engs, vals)]
        g.all_reduce_sum(bufs, 1000)
        g.sync()
        want = vals[0].copy()
        for v in vals[1:]:
            want = want + v          # the same left-to-right f32 adds
        for b in bufs:
            assert np.array_equal(b.download((1000,), np.float32), want)
        h = [np.arange(16, dtype=np.uint64) * (i + 1) + (1 << 40) for i in range(members)]
        hb = [e.to_device(v) for e, v in zip(engs, h)]
        g.all_reduce_u64(hb, 16)
        g.sync()
        assert np.array_equal(hb[-1].download((16,), np.uint64), sum(h))
        counts = [5 + 3 * i for i in range(members)]
        send = [e.to_device(np.full(c, i + 1, np.float32)) for i, (e, c) in enumerate(zip(engs, counts))]
        recv = engs[0].empty((sum(counts),))
        g.gather(send, counts, recv)
        g.sync()
        assert np.array_equal(recv.download((sum(counts),), np.float32),
                              np.concatenate([np.full(c, i + 1, np.float32) for i, c in enumerate(counts)]))
        for b in bufs + hb + send + [recv]:
            b.free()


@pytest.mark.parametrize("shape,members", [((7, 6, 1024), 2), ((9, 4, 4096), 3), ((5, 8, 1001), 2), ((4, 4, 256), 1)])
def test_group_session_matches_single_session(engine, shape, members):
    """slabs recomputed side by side + C2 + C1 == one session over the whole cube: per-pixel outputs bit for bit
    (the same kernels on the same traces), pixel means to 1e-6 (sums associate differently), all gather levels"""
    nx, ny, nt = shape
    time, cube = synth.make_cube(nx, ny, nt)
    cfg = pkg.chain_cfg_default(time)
    single = pkg.Session(engine, nx, ny, time)
    try:
        single.upload(cube, subtract_bias=False)
        single.recompute(cfg)
        want = {w: single.download(w) for w in (pkg.BUF_IMG, pkg.BUF_DATA, pkg.BUF_FFT, pkg.BUF_AMPLITUDES, pkg.BUF_PHASES,
                                                pkg.BUF_AVG_FFT, pkg.BUF_AVG_AMPLITUDES, pkg.BUF_AVG_PHASES)}
    finally:
        single.close()
    with pkg.Group(devices=[0] * members) as g:
        for q in range(members):
            assert pkg.host_slab(nx, members, q) == __import__("thz_image_explorer_amd.shard", fromlist=["slab"]).slab(nx, members, q)
        gs = pkg.GroupSession(g, nx, ny, time)
        try:
            gs.upload(cube, subtract_bias=False)
            for level, bufs in ((pkg.GATHER_SMALL, (pkg.BUF_IMG,)), (pkg.GATHER_TIME, (pkg.BUF_IMG, pkg.BUF_DATA)),
                                (pkg.GATHER_ALL, (pkg.BUF_IMG, pkg.BUF_DATA, pkg.BUF_FFT, pkg.BUF_AMPLITUDES, pkg.BUF_PHASES))):
                gs.recompute(cfg, 1, level)
                for w in bufs:
                    assert np.array_equal(gs.download(w), want[w]), f"gather level {level}, buffer {w}"
                for w in (pkg.BUF_DATA, pkg.BUF_FFT):
                    if w not in bufs:
                        with pytest.raises(pkg.ThzError):
                            gs.download(w)
                for w in (pkg.BUF_AVG_FFT, pkg.BUF_AVG_AMPLITUDES, pkg.BUF_AVG_PHASES):
                    assert rel(gs.download(w), want[w]) < 2e-6
            # a Time Band Pass (after) slider: chain position 7 only re-runs the tail on every slab
            cfg2 = pkg.chain_cfg_default(time)
            cfg2.td_after_low = float(time[0]) + 3.0
            gs.recompute(cfg2, 7, pkg.GATHER_TIME)
            ref = ob.run_pipeline(cube, time, synth.oracle_chain(time))
            w2 = ob.td_bandpass_window(time, cfg2.td_after_low, cfg2.td_after_high, cfg2.td_after_width)[0]
            w1 = synth.oracle_chain(time)["w_post"]
            sel = w1 == 1.0       # where the default taper is 1 the new result is ref * w2
            got = gs.download(pkg.BUF_DATA).reshape(nx, ny, nt)
            assert rel(got[..., sel], (ref["data"] * w2)[..., sel], np.abs(ref["data"]).max()) < TOL
        finally:
            gs.close()


def test_rank_api_through_real_rccl_single_rank(engine, monkeypatch):
    """The process-per-GPU entry points (thz_group_unique_id + thz_group_create_rank — what bench.py --gpus N and the
    Rust data threads call) with THZ_GROUP_FORCE_RCCL: librccl is dlopen'ed, the communicator is built with
    ncclCommInitRank and the collectives of upload / recompute / deconvolve go through ncclAllReduce, grouped
    ncclSend / ncclRecv and ncclBroadcast ON REAL RCCL — with the one rank a one-GPU box can hold.  What stays
    untested here is only the fabric."""
    monkeypatch.setenv("THZ_GROUP_FORCE_RCCL", "1")
    nx, ny, nt = 6, 5, 1024
    time, cube = synth.make_cube(nx, ny, nt)
    cfg = pkg.chain_cfg_default(time)
    single = pkg.Session(engine, nx, ny, time)
    try:
        single.upload(cube, subtract_bias=False)
        single.recompute(cfg)
        bufs = (pkg.BUF_IMG, pkg.BUF_DATA, pkg.BUF_FFT, pkg.BUF_AMPLITUDES, pkg.BUF_PHASES)
        means = (pkg.BUF_AVG_FFT, pkg.BUF_AVG_AMPLITUDES, pkg.BUF_AVG_PHASES)
        want = {w: single.download(w) for w in bufs + means}
    finally:
        single.close()
    uid = pkg.group_unique_id()
    assert len(uid) == 128 and any(uid)
    with pkg.Group(device=0, rank=0, world=1, uid=uid) as g:
        assert g.world == 1 and g.ranks == [0]
        e = g.engine(0)
        v = np.arange(1000, dtype=np.float32)
        b = e.to_device(v)
        g.all_reduce_sum([b], 1000)          # ncclAllReduce over one rank: the values come back unchanged
        g.sync()
        assert np.array_equal(b.download((1000,), np.float32), v)
        h = e.to_device(np.arange(16, dtype=np.uint64) + (1 << 40))
        g.all_reduce_u64([h], 16)
        g.sync()
        assert np.array_equal(h.download((16,), np.uint64), np.arange(16, dtype=np.uint64) + (1 << 40))
        gs = pkg.GroupSession(g, nx, ny, time)
        try:
            gs.upload(cube, subtract_bias=False)
            gs.recompute(cfg, 1, pkg.GATHER_ALL)
            for w in bufs:
                assert np.array_equal(gs.download(w), want[w])
            for w in means:
                assert rel(gs.download(w), want[w]) < 2e-6
        finally:
            gs.close()
        b.free(); h.free()
        # the band-parallel Deconvolution stage: all-gather of the slabs by ncclBroadcast, all-reduce of the band sums
        import os
        nx2, ny2, nt2 = 36, 32, 256
        time2, cube2 = synth.make_cube(nx2, ny2, nt2)
        psf = pkg.psf_from_npz(np.load(os.path.join(os.path.dirname(__file__), "golden", "psf_sample.npz")))
        cfg2, dcfg = pkg.chain_cfg_default(time2), pkg.DeconvCfg(20, 5, 0.4, 3.0, 0.5)
        single = pkg.Session(engine, nx2, ny2, time2, 0.5, 0.5)
        try:
            single.upload(cube2, subtract_bias=False)
            single.recompute(cfg2)
            assert single.deconvolve(psf, dcfg) == 0
            want_d, want_i = single.download(pkg.BUF_DATA), single.download(pkg.BUF_IMG)
        finally:
            single.close()
        gs = pkg.GroupSession(g, nx2, ny2, time2, 0.5, 0.5)
        try:
            gs.upload(cube2, subtract_bias=False)
            gs.recompute(cfg2, 1, pkg.GATHER_TIME)
            assert gs.deconvolve(psf, dcfg) == 0
            assert rel(gs.download(pkg.BUF_DATA), want_d) < TOL
            assert rel(gs.download(pkg.BUF_IMG), want_i) < TOL
        finally:
            gs.close()


def _variant_cfg(time, variant):
    cfg = pkg.chain_cfg_default(time)
    if variant == "scale2+tilt+means2":
        cfg.scale_factor, cfg.tilt_x_deg, cfg.want_means = 2, 1.5, 2
    elif variant.startswith("scale"):
        cfg.scale_factor = int(variant[5:])
    elif variant == "tilt":
        cfg.tilt_x_deg, cfg.tilt_y_deg = 2.0, -1.0
    elif variant == "means2":
        cfg.want_means = 2
    return cfg


@pytest.mark.parametrize("variant", ["scale2", "scale3", "tilt", "means2", "scale2+tilt+means2"])
@pytest.mark.parametrize("shape,members", [((13, 6, 256), 2), ((17, 8, 1024), 3), ((12, 6, 1001), 4)])
def test_group_session_shards_what_it_used_to_refuse(engine, shape, members, variant):
    """Round 2's group session refused scale_factor > 1, a non-zero tilt and want_means = 2 (THZ_ERR_UNSUPPORTED).
    Now: block means over slab edges continue the previous slab's partial sums (a block belongs to the slab that
    holds its last row), the Tilt plan is made for the whole grid, and the reference-order means run slab after slab
    on a carried running sum — per-pixel outputs and the reference-order means bit for bit one session's."""
    nx, ny, nt = shape
    time, cube = synth.make_cube(nx, ny, nt)
    cfg = _variant_cfg(time, variant)
    poly = np.array([[1, 1], [4, 1], [5, 6], [2, 9], [0, 5]], np.uint64)
    bufs = (pkg.BUF_IMG, pkg.BUF_DATA, pkg.BUF_FFT, pkg.BUF_AMPLITUDES, pkg.BUF_PHASES)
    avgs = (pkg.BUF_AVG_FFT, pkg.BUF_AVG_AMPLITUDES, pkg.BUF_AVG_PHASES)
    single = pkg.Session(engine, nx, ny, time, 0.5, 0.5)
    try:
        single.upload(cube, subtract_bias=False)
        single.set_rois([poly])
        single.recompute(cfg)
        nto = single.nt_out
        want = {w: single.download(w) for w in bufs + avgs}
        want_roi = single.roi(0)
        px = nx // 2
        want_plot = single.plot(px, 1)
        grid = single.grid()
    finally:
        single.close()
    if members > 1 and "scale" in variant and min(pkg.host_slab(nx, members, q)[1] for q in range(members)) < cfg.scale_factor:
        pytest.skip("a slab shorter than the scale factor is refused (a block would span three slabs)")
    with pkg.Group(devices=[0] * members) as g:
        gs = pkg.GroupSession(g, nx, ny, time, 0.5, 0.5)
        try:
            gs.upload(cube, subtract_bias=False)
            gs.set_rois([poly])
            gs.recompute(cfg, 1, pkg.GATHER_ALL)
            # lengths that are not a power of two (1001; any tilted cube) are transformed in PAIRS of traces: a slab that
            # starts at an odd trace pairs them differently than one session does, and the last bits may differ
            pairs = nto & (nto - 1) != 0
            for w in bufs:
                got = gs.download(w, nt_out=nto)
                assert got.shape == want[w].shape, (w, got.shape, want[w].shape, grid)
                if w == pkg.BUF_PHASES and pairs:
                    d = got.astype(np.float64) - want[w]
                    assert np.abs(d - 2 * np.pi * np.round(d / (2 * np.pi))).max() < 3e-3
                elif pairs:
                    assert rel(got, want[w]) < 2e-6, w
                else:
                    assert np.array_equal(got, want[w]), w
            for w in avgs:
                got = gs.download(w, nt_out=nto)
                if cfg.want_means == 2 and not pairs:
                    assert np.array_equal(got, want[w]), w          # the reference's order, slab after slab
                else:
                    assert rel(got, want[w]) < 2e-6 or (w == pkg.BUF_AVG_PHASES and pairs), w
            r = gs.roi(0, nt_out=nto)
            assert r["count"] == want_roi["count"]
            for k in ("signal_fft", "signal", "roi_data"):
                assert np.abs(r[k].astype(np.float64) - want_roi[k]).max() <= 2e-6 * max(np.abs(want_roi[k]).max(), 1e-30), k
            # the plot copy-out of a pixel: raw trace from the slab that holds the row, processed traces from the slab
            # that holds its block
            x0s = [pkg.host_slab(nx, members, q)[0] for q in range(members)]
            got_plot = None
            for i in range(members):
                n_i = pkg.host_slab(nx, members, i)[1]
                if x0s[i] <= px < x0s[i] + n_i:
                    try:
                        got_plot = gs.member(i).plot(px - x0s[i], 1, want=["filtered_signal", "filtered_signal_fft"])
                    except pkg.ThzError:
                        got_plot = None       # the block is the next slab's
            if got_plot is not None and not pairs:
                assert np.array_equal(got_plot["filtered_signal"], want_plot["filtered_signal"])
        finally:
            gs.close()
    with pytest.raises(pkg.ThzError):
        pkg.Group(devices=[0, 0, 1])   # neither all the same nor all distinct


def test_group_session_band_parallel_deconvolution(engine):
    """BASELINE config 4's "1 -> 2 GPUs": the Deconvolution stage over the group — every member the transform, band
    energies and recombination of its own rows, the Richardson-Lucy iterations of its own bands over the whole image,
    energies and gains exchanged as 2-D images (round 2: all-gather of the cube, all-reduce of band sums) — equals the
    single-session stage; a guard on every rank and an abort both leave the input as the stage's output"""
    import ctypes
    import os
    nx, ny, nt = 36, 32, 256
    time, cube = synth.make_cube(nx, ny, nt)
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "psf_sample.npz"))
    psf = pkg.psf_from_npz(z)
    cfg = pkg.chain_cfg_default(time)
    dcfg = pkg.DeconvCfg(20, 5, 0.4, 3.0, 0.5)
    single = pkg.Session(engine, nx, ny, time, 0.5, 0.5)
    try:
        single.upload(cube, subtract_bias=False)
        single.recompute(cfg)
        plain = single.download(pkg.BUF_DATA).copy()
        assert single.deconvolve(psf, dcfg) == 0
        want, want_img = single.download(pkg.BUF_DATA), single.download(pkg.BUF_IMG)
    finally:
        single.close()
    for members in (2, 3):
        with pkg.Group(devices=[0] * members) as g:
            gs = pkg.GroupSession(g, nx, ny, time, 0.5, 0.5)
            try:
                gs.upload(cube, subtract_bias=False)
                gs.recompute(cfg, 1, pkg.GATHER_TIME)
                assert np.array_equal(gs.download(pkg.BUF_DATA), plain)
                img_plain = gs.download(pkg.BUF_IMG).copy()
                assert gs.deconvolve(psf, dcfg) == 0
                got, img = gs.download(pkg.BUF_DATA), gs.download(pkg.BUF_IMG)
                # every pixel's recombination runs over all bands in one member, in the single session's order: what can
                # differ is which traces share a transform — nothing here (nt = 256)
                assert rel(got, want) < 1e-6
                assert rel(img, want_img) < 1e-6
                # slabs hold their rows of the same result
                rows = np.concatenate([gs.member(i).download(pkg.BUF_DATA) for i in range(members)])
                assert np.array_equal(rows, got)
                # an abort seen by the ranks: every rank passes its input through, the group returns ABORTED
                abort = ctypes.c_int(1)
                with pytest.raises(pkg.ThzError) as e:
                    gs.deconvolve(psf, pkg.DeconvCfg(80, 5, 0.4, 3.0, 0.5), abort=abort)
                assert e.value.code == -5
                assert np.array_equal(gs.download(pkg.BUF_IMG), img_plain)
                assert np.array_equal(gs.download(pkg.BUF_DATA), plain)
                # a guard (n_filters < 2) on every rank: skipped, input unchanged
                gs.recompute(cfg, 1, pkg.GATHER_TIME)
                assert gs.deconvolve(psf, pkg.DeconvCfg(20, 1, 0.4, 3.0, 0.5)) == 1
                assert np.array_equal(gs.download(pkg.BUF_DATA), plain)
            finally:
                gs.close()
