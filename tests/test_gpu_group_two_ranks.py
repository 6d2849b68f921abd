"""Two (and three) RANKS, one process each, through the library's process-per-GPU entry points — on ONE GPU.

RCCL refuses two ranks on one device, so this test swaps RCCL for tests/mock_rccl (THZ_RCCL_LIB: the eleven entry
points group_api.cpp resolves, with their real signatures, over shared memory and a process-shared barrier; test
infrastructure, see its header).  Everything else is the real thing: thz_group_unique_id / thz_group_create_rank, one
process per rank with its own HIP context, thz_group_session_upload / _recompute / _deconvolve / _download — the
call sequence bench.py --gpus N and the Rust data threads make.  What it pins: who sends what to whom (grouped
ncclSend / ncclRecv of the gather), counts and offsets per rank for ragged slabs, root and non-root roles, the
all-reduce of the pixel sums on every rank, the broadcast-built all-gather and the band split of the deconvolution.
What it cannot say anything about is the fabric."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
MOCK_DIR = os.path.join(HERE, "mock_rccl")
MOCK = os.path.join(MOCK_DIR, "librccl_mock.so")


def _build_mock():
    src = os.path.join(MOCK_DIR, "mock_rccl.cpp")
    if os.path.exists(MOCK) and os.path.getmtime(MOCK) >= os.path.getmtime(src):
        return
    subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src, "-o", MOCK,
                    "-L/opt/rocm/lib", "-lamdhip64", "-lpthread", "-lrt"], check=True)


RANK_SCRIPT = textwrap.dedent('''
    import os, sys, time
    sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
    import numpy as np
    import thz_image_explorer_amd as pkg
    import synth
    rank, world, uid_file, out_file = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    nx, ny, nt = {shape!r}
    if rank == 0:
        uid = pkg.group_unique_id()
        with open(uid_file + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(uid_file + ".tmp", uid_file)
    else:
        for _ in range(3000):
            if os.path.exists(uid_file):
                break
            time.sleep(0.01)
        uid = open(uid_file, "rb").read()
    time_axis, cube = synth.make_cube(nx, ny, nt)
    res = {{}}
    with pkg.Group(device=0, rank=rank, world=world, uid=uid) as g:
        assert g.world == world and g.ranks == [rank]
        gs = pkg.GroupSession(g, nx, ny, time_axis, 0.5, 0.5)
        try:
            gs.upload(cube, subtract_bias=False)          # every rank takes its rows of the one cube
            cfg = pkg.chain_cfg_default(time_axis)
            for level in (pkg.GATHER_SMALL, pkg.GATHER_ALL):
                gs.recompute(cfg, 1, level)
            # every rank holds the cube's means (C2 is an all-reduce)
            sess = gs.member(0)
            res["avg_amp_rank%d" % rank] = sess.download(pkg.BUF_AVG_AMPLITUDES)
            res["avg_ph_rank%d" % rank] = sess.download(pkg.BUF_AVG_PHASES)
            res["avg_fft_rank%d" % rank] = sess.download(pkg.BUF_AVG_FFT)
            if rank == 0:
                for name, w in (("img", pkg.BUF_IMG), ("data", pkg.BUF_DATA), ("fft", pkg.BUF_FFT), ("amp", pkg.BUF_AMPLITUDES),
                                ("ph", pkg.BUF_PHASES)):
                    res[name] = gs.download(w)
            if {deconv!r}:
                psf = pkg.psf_from_npz(np.load(os.path.join({tests!r}, "golden", "psf_sample.npz")))
                st = gs.deconvolve(psf, pkg.DeconvCfg(20, 5, 0.4, 3.0, 0.5))
                res["deconv_status_rank%d" % rank] = np.array([st])
                if rank == 0:
                    res["deconv_data"] = gs.download(pkg.BUF_DATA)
                    res["deconv_img"] = gs.download(pkg.BUF_IMG)
        finally:
            gs.close()
    np.savez(out_file, **res)
''')


def pkg_slab_start(nx, world, rank):
    import thz_image_explorer_amd as pkg
    return pkg.host_slab(nx, world, rank)[0]


def _single_session(engine, shape, deconv):
    import synth
    import thz_image_explorer_amd as pkg
    nx, ny, nt = shape
    time_axis, cube = synth.make_cube(nx, ny, nt)
    s = pkg.Session(engine, nx, ny, time_axis, 0.5, 0.5)
    try:
        s.upload(cube, subtract_bias=False)
        s.recompute(pkg.chain_cfg_default(time_axis))
        want = {n: s.download(w) for n, w in (("img", pkg.BUF_IMG), ("data", pkg.BUF_DATA), ("fft", pkg.BUF_FFT), ("amp", pkg.BUF_AMPLITUDES),
                                              ("ph", pkg.BUF_PHASES), ("avg_amp", pkg.BUF_AVG_AMPLITUDES),
                                              ("avg_ph", pkg.BUF_AVG_PHASES), ("avg_fft", pkg.BUF_AVG_FFT))}
        if deconv:
            psf = pkg.psf_from_npz(np.load(os.path.join(HERE, "golden", "psf_sample.npz")))
            assert s.deconvolve(psf, pkg.DeconvCfg(20, 5, 0.4, 3.0, 0.5)) == 0
            want["deconv_data"], want["deconv_img"] = s.download(pkg.BUF_DATA), s.download(pkg.BUF_IMG)
    finally:
        s.close()
    return want


@pytest.mark.parametrize("world,shape,deconv", [(2, (7, 6, 1024), False), (3, (8, 5, 1001), False), (2, (36, 32, 256), True)])
def test_rank_processes_match_single_session(engine, tmp_path, world, shape, deconv):
    from test_gpu_parity import TOL, rel
    _build_mock()
    want = _single_session(engine, shape, deconv)
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT.format(root=ROOT, tests=HERE, shape=shape, deconv=deconv))
    uid_file = str(tmp_path / "uid.bin")
    env = dict(os.environ, THZ_RCCL_LIB=MOCK, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), uid_file, str(tmp_path / f"out{r}.npz")], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()          # the exact children started above
            pytest.fail("a rank process did not finish: the ranks' calls do not pair up")
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}:\n{o[-3000:]}"
    res = [np.load(str(tmp_path / f"out{r}.npz")) for r in range(world)]
    # per-pixel outputs gathered on rank 0: the same kernels on the same traces, bit for bit — unless the length's
    # kernels transform traces in PAIRS (nt = 1001: x1 + i x2 through one complex transform) and a slab starts at an
    # odd trace: then other traces share a transform than in the single session, and the last bits may differ
    nx, ny, nt = shape
    pairs_shift = nt & (nt - 1) != 0 and any((pkg_slab_start(nx, world, r) * ny) % 2 for r in range(world))
    for name in ("img", "data", "fft", "amp"):
        if pairs_shift:
            assert rel(res[0][name], want[name]) < 2e-6, name
        else:
            assert np.array_equal(res[0][name], want[name]), name
    if pairs_shift:
        d = res[0]["ph"].astype(np.float64) - want["ph"]
        assert np.abs(d - 2 * np.pi * np.round(d / (2 * np.pi))).max() < 3e-3   # last-bit inputs may flip a 2 pi decision on a noise bin
    else:
        assert np.array_equal(res[0]["ph"], want["ph"])
    # pixel means: present and equal on EVERY rank; sums of slab sums associate differently than one session's
    for r in range(world):
        for name in ("avg_amp", "avg_ph", "avg_fft"):
            got = res[r][f"{name}_rank{r}"]
            assert rel(got, want[name]) < 2e-6, (name, r)
            assert np.array_equal(got, res[0][f"{name}_rank0"]), (name, r)
    if deconv:
        for r in range(world):
            assert int(res[r][f"deconv_status_rank{r}"][0]) == 0
        assert rel(res[0]["deconv_data"], want["deconv_data"]) < TOL     # the band sums associate differently
        assert rel(res[0]["deconv_img"], want["deconv_img"]) < TOL


HALO_SCRIPT = textwrap.dedent('''
    import os, sys, time
    sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
    import numpy as np
    import thz_image_explorer_amd as pkg
    import synth
    from test_gpu_group import _variant_cfg
    rank, world, uid_file, out_file = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    nx, ny, nt = {shape!r}
    if rank == 0:
        uid = pkg.group_unique_id()
        with open(uid_file + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(uid_file + ".tmp", uid_file)
    else:
        for _ in range(3000):
            if os.path.exists(uid_file):
                break
            time.sleep(0.01)
        uid = open(uid_file, "rb").read()
    time_axis, cube = synth.make_cube(nx, ny, nt)
    res = {{}}
    with pkg.Group(device=0, rank=rank, world=world, uid=uid) as g:
        gs = pkg.GroupSession(g, nx, ny, time_axis, 0.5, 0.5)
        try:
            gs.upload(cube, subtract_bias=False)
            gs.set_rois([np.array([[1, 1], [4, 1], [5, 6], [2, 9], [0, 5]], np.uint64)])
            gs.recompute(_variant_cfg(time_axis, {variant!r}), 1, pkg.GATHER_ALL)
            nto = gs.member(0).nt_out
            sess = gs.member(0)
            for name, w in (("avg_amp", pkg.BUF_AVG_AMPLITUDES), ("avg_ph", pkg.BUF_AVG_PHASES), ("avg_fft", pkg.BUF_AVG_FFT)):
                res[name + "_rank%d" % rank] = sess.download(w)
            r = gs.roi(0, nt_out=nto)
            res["roi_amp_rank%d" % rank] = r["signal_fft"]
            res["roi_count_rank%d" % rank] = np.array([r["count"]])
            if rank == 0:
                for name, w in (("img", pkg.BUF_IMG), ("data", pkg.BUF_DATA), ("fft", pkg.BUF_FFT), ("amp", pkg.BUF_AMPLITUDES)):
                    res[name] = gs.download(w, nt_out=nto)
        finally:
            gs.close()
    np.savez(out_file, **res)
''')


@pytest.mark.parametrize("world,shape,variant", [(2, (13, 6, 256), "scale2"), (3, (17, 8, 1024), "scale3"), (3, (17, 8, 1024), "means2"),
                                                 (2, (13, 6, 256), "tilt"), (3, (13, 6, 256), "scale2+tilt+means2")])
def test_rank_processes_halo_paths(engine, tmp_path, world, shape, variant):
    """what the group session refused until round 3, with one PROCESS per rank: the partial block sums and the carried
    running sums of the reference-order means travel rank q -> q + 1 as ncclSend / ncclRecv pairs while the other ranks
    are elsewhere (the mock's point-to-point is pairwise for this), the Tilt plan is the whole grid's on every rank"""
    import synth
    import thz_image_explorer_amd as pkg
    from test_gpu_group import _variant_cfg
    from test_gpu_parity import rel
    _build_mock()
    nx, ny, nt = shape
    time_axis, cube = synth.make_cube(nx, ny, nt)
    cfg = _variant_cfg(time_axis, variant)
    s = pkg.Session(engine, nx, ny, time_axis, 0.5, 0.5)
    try:
        s.upload(cube, subtract_bias=False)
        s.set_rois([np.array([[1, 1], [4, 1], [5, 6], [2, 9], [0, 5]], np.uint64)])
        s.recompute(cfg)
        nto = s.nt_out
        want = {n: s.download(w) for n, w in (("img", pkg.BUF_IMG), ("data", pkg.BUF_DATA), ("fft", pkg.BUF_FFT), ("amp", pkg.BUF_AMPLITUDES),
                                              ("avg_amp", pkg.BUF_AVG_AMPLITUDES), ("avg_ph", pkg.BUF_AVG_PHASES), ("avg_fft", pkg.BUF_AVG_FFT))}
        want_roi = s.roi(0)
    finally:
        s.close()
    script = tmp_path / "rank.py"
    script.write_text(HALO_SCRIPT.format(root=ROOT, tests=HERE, shape=shape, variant=variant))
    uid_file = str(tmp_path / "uid.bin")
    env = dict(os.environ, THZ_RCCL_LIB=MOCK, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), uid_file, str(tmp_path / f"out{r}.npz")], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()          # the exact children started above
            pytest.fail("a rank process did not finish: the ranks' calls do not pair up")
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}:\n{o[-3000:]}"
    res = [np.load(str(tmp_path / f"out{r}.npz")) for r in range(world)]
    pairs = nto & (nto - 1) != 0
    for name in ("img", "data", "fft", "amp"):
        assert res[0][name].shape == want[name].shape, name
        if pairs:
            assert rel(res[0][name], want[name]) < 2e-6, name
        else:
            assert np.array_equal(res[0][name], want[name]), name
    for r in range(world):
        for name in ("avg_amp", "avg_ph", "avg_fft"):
            got = res[r][f"{name}_rank{r}"]
            if cfg.want_means == 2 and not pairs:
                assert np.array_equal(got, want[name]), (name, r)      # the reference's sequential order, rank after rank
            elif not (name == "avg_ph" and pairs):
                assert rel(got, want[name]) < 2e-6, (name, r)
            assert np.array_equal(got, res[0][f"{name}_rank0"]), (name, r)
        assert int(res[r][f"roi_count_rank{r}"][0]) == want_roi["count"]
        assert rel(res[r][f"roi_amp_rank{r}"], want_roi["signal_fft"]) < 2e-6
