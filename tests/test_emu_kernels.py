"""Index arithmetic of the HIP kernels, checked on the CPU: the product's
csrc/kernels.hip is compiled with -DTHZ_EMU (every lane a host thread, see
tests/emu/hip_emu.h) and compared with the oracle.  This does not replace the
`-m gpu` parity tests; it catches lane/bin mapping and LDS exchange bugs in a
container that has no GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_binding as ob
import synth

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_SO = os.path.join(HERE, "emu", "libthz_emu.so")
_P = C.c_void_p


@pytest.fixture(scope="module")
def emu():
    import glob
    csrc = os.path.join(HERE, "..", "thz_image_explorer_amd", "csrc")
    srcs = glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp")) \
        + glob.glob(os.path.join(HERE, "emu", "*.cpp")) + glob.glob(os.path.join(HERE, "emu", "*.h"))
    if (not os.path.exists(EMU_SO)) or os.path.getmtime(EMU_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["bash", os.path.join(HERE, "emu", "build_emu.sh")])
    return C.CDLL(EMU_SO)


def _p(a):
    return None if a is None else a.ctypes.data_as(_P)


def _fwd(emu, x, wa, mask, nt):
    npix = x.shape[0]
    nf = nt // 2 + 1
    dout = np.zeros_like(x)
    fft = np.zeros((npix, nf, 2), np.float32)
    amp = np.zeros((npix, nf), np.float32)
    ph = np.zeros((npix, nf), np.float32)
    rc = emu.emu_fft_fwd(nt, C.c_size_t(npix), _p(x), _p(wa), None, _p(dout), _p(fft), _p(amp), _p(ph), _p(mask))
    assert rc == 0
    return dout, fft, amp, ph


P_LENGTHS = (1001, 1000, 1200, 1500, 2000)   # trace lengths with a mixed-radix (P) kernel, plan_host.hpp p_factors


@pytest.mark.parametrize("family", ["auto", "g", "nop"])
@pytest.mark.parametrize("nt", [4, 16, 64, 256, 1024, 2048, 4096, 1001, 1000, 30, 1500, 3000])
def test_forward_inverse_vs_oracle(emu, nt, family):
    emu.emu_allow_f(0 if family == "g" else 1)
    emu.emu_allow_p(1 if family == "auto" else 0)
    if family == "g" and nt not in (1024, 2048, 4096):
        pytest.skip("only one family exists for this length")
    if family == "nop" and nt not in P_LENGTHS:
        pytest.skip("no P kernel for this length anyway")
    # auto: F for 1024/2048/4096, P (mixed radix) for 1001 / 1000, FB / FB2 / FB4 (chirp-z over the F core) for
    # the other lengths below 4096 that are not a power of two; nop: the same without the P kernels
    want = 0
    if family != "g":
        want = 1 if nt in (1024, 2048, 4096) else ((2 if nt < 1024 else 3 if nt < 2048 else 4) if nt & (nt - 1) else 0)
        if family == "auto" and nt in P_LENGTHS:
            want = 6
    assert emu.emu_family(nt) == want
    npix = 11  # > waves per block: exercises the grid-stride loop and a ragged last block
    rng = np.random.default_rng(nt)
    time = synth.make_time(nt)
    x = synth.make_traces(np.arange(npix) + 7, nt) + 0.05 * rng.standard_normal((npix, nt)).astype(np.float32)
    x = np.ascontiguousarray(x, np.float32)
    wa = ob.apply_window(0, np.ones(nt, np.float32), time, 1.0, 7.0) if nt >= 256 else np.ones(nt, np.float32)
    freq = ob.frequency_axis(time)
    mask = ob.fd_bandpass_window(freq, 0.2, 5.0, 0.1)[0] if nt >= 256 else np.ones(nt // 2 + 1, np.float32)
    dout, fft, amp, ph = _fwd(emu, x, wa, mask, nt)
    st = ob.fft_stage((x * wa).reshape(1, npix, nt), time, 0, 0.0, 0.0)
    scale = np.abs(st["fft"]).max()
    assert np.array_equal(dout, x * wa)
    ref_f, ref_a = st["fft"][0] * mask[None, :, None], st["amplitudes"][0] * mask[None, :]
    assert np.abs(fft - ref_f).max() / scale < 1e-5
    assert np.abs(amp - ref_a).max() / scale < 1e-5
    d = ph - st["phases"][0]
    assert np.abs(d - 2 * np.pi * np.round(d / (2 * np.pi))).max() < 2e-3
    out = np.zeros_like(x)
    img = np.zeros(npix, np.float32)
    win = ob.td_bandpass_window(time, float(time[0]), float(time[-1]), 0.1)[0]
    assert emu.emu_fft_inv(nt, C.c_size_t(npix), _p(np.ascontiguousarray(ref_f)), _p(win), _p(out), _p(img)) == 0
    back, _ = ob.ifft_stage(ref_f.reshape(1, npix, -1, 2), nt)
    ref_t = back[0] * win
    assert np.abs(out - ref_t).max() / np.abs(ref_t).max() < 1e-5
    assert np.abs(img - (ref_t.astype(np.float64) ** 2).sum(1)).max() / img.max() < 1e-5
    emu.emu_allow_p(1)


@pytest.mark.parametrize("nt", [256, 1024, 2048, 4096])
def test_fused_pipeline_vs_oracle(emu, nt):
    emu.emu_allow_f(1)
    nx, ny = 3, 5
    time, cube = synth.make_cube(nx, ny, nt)
    chain = synth.default_chain(time)
    npix = nx * ny
    nf = nt // 2 + 1
    fft = np.zeros((npix, nf, 2), np.float32); amp = np.zeros((npix, nf), np.float32)
    ph = np.zeros((npix, nf), np.float32); out = np.zeros((npix, nt), np.float32); img = np.zeros(npix, np.float32)
    rc = emu.emu_pipeline(nt, C.c_size_t(npix), _p(cube), _p(chain["w_pre"]), _p(chain["fd_mask"]),
                          _p(chain["w_post"]), _p(fft), _p(amp), _p(ph), _p(out), _p(img))
    assert rc == 0
    ref = ob.run_pipeline(cube, time, chain)
    scale = np.abs(ref["fft"]).max()
    assert np.abs(fft.reshape(ref["fft"].shape) - ref["fft"]).max() / scale < 1e-5
    assert np.abs(amp.reshape(ref["amplitudes"].shape) - ref["amplitudes"]).max() / scale < 1e-5
    assert np.abs(out.reshape(ref["data"].shape) - ref["data"]).max() / np.abs(ref["data"]).max() < 1e-5
    assert np.abs(img.reshape(ref["img"].shape) - ref["img"]).max() / ref["img"].max() < 1e-5


@pytest.mark.parametrize("nt", [1001, 1000, 1024 - 1, 513, 512 - 1, 300, 257, 129, 40, 1025, 1500, 2000, 2047, 2049, 3000, 4000, 4095, 4097, 6000, 8191])
def test_chirpz_pipeline_vs_oracle(emu, nt):
    """FB / FB2 kernels (fft_fb.hpp): non-power-of-two trace lengths, chirp-z over the F core
    (one core run per transform up to nt = 1023, two up to 2047, four up to 4095, eight up to 8191)"""
    emu.emu_allow_f(1)
    emu.emu_allow_p(0)   # 1001 / 1000 have a mixed-radix kernel of their own (test_mixed_radix_pipeline_vs_oracle)
    assert emu.emu_family(nt) == (2 if nt < 1024 else 3 if nt < 2048 else 4 if nt < 4096 else 5)
    nx, ny = (5, 1) if nt % 2 else (2, 3)   # odd trace count: the last pair has one member
    time = synth.make_time(nt)
    cube = synth.make_traces(np.arange(nx * ny) + 11, max(nt, 320))[:, :nt].reshape(nx, ny, nt).copy()
    chain = synth.default_chain(time)
    npix, nf = nx * ny, nt // 2 + 1
    fft = np.zeros((npix, nf, 2), np.float32); amp = np.zeros((npix, nf), np.float32)
    ph = np.zeros((npix, nf), np.float32); out = np.zeros((npix, nt), np.float32); img = np.zeros(npix, np.float32)
    rc = emu.emu_pipeline(nt, C.c_size_t(npix), _p(cube), _p(chain["w_pre"]), _p(chain["fd_mask"]),
                          _p(chain["w_post"]), _p(fft), _p(amp), _p(ph), _p(out), _p(img))
    assert rc == 0
    ref = ob.run_pipeline(cube, time, chain)
    scale = np.abs(ref["fft"]).max()
    assert np.abs(fft.reshape(ref["fft"].shape) - ref["fft"]).max() / scale < 1e-5
    assert np.abs(amp.reshape(ref["amplitudes"].shape) - ref["amplitudes"]).max() / scale < 1e-5
    assert np.abs(out.reshape(ref["data"].shape) - ref["data"]).max() / max(np.abs(ref["data"]).max(), 1e-30) < 1e-5
    assert np.abs(img.reshape(ref["img"].shape) - ref["img"]).max() / max(ref["img"].max(), 1e-30) < 1e-5
    # unwrapped phases on the strong bins of the unmasked spectrum
    st = ob.fft_stage((cube * chain["w_pre"]).astype(np.float32), time, 0, 0.0, 0.0)
    strong = st["amplitudes"] > 0.05 * st["amplitudes"].max(axis=-1, keepdims=True)
    d = ph.reshape(ref["phases"].shape).astype(np.float64) - ref["phases"]
    assert np.abs(d - 2 * np.pi * np.round(d / (2 * np.pi)))[strong].max() < 3e-3


@pytest.mark.parametrize("pairs", [1, 2])
@pytest.mark.parametrize("nt,npix_shape", [(1001, (5, 1)), (1001, (2, 3)), (1001, (1, 1)), (1001, (7, 1)), (1000, (5, 1)), (1000, (3, 6)),
                                           (1200, (5, 1)), (1500, (3, 1)), (2000, (3, 1))])
def test_mixed_radix_pipeline_vs_oracle(emu, nt, npix_shape, pairs):
    """P kernels (fft_p.hpp): nt = 1001 = 7 x 11 x 13 (the length of real scans) and 1000 = 10 x 10 x 10 as one
    direct three-pass mixed-radix transform per pair of traces; odd and even trace counts, more pairs than one
    block has waves"""
    emu.emu_allow_f(1)
    emu.emu_allow_p(1)
    if pairs == 2 and nt > 1001:
        pytest.skip("the round lengths 1200 / 1500 / 2000 are built with one pair per wave only")
    emu.emu_set_p_pairs(pairs)   # one or two pairs of traces per wave: units of 2 or 4 traces, ragged last unit
    assert emu.emu_family(nt) == 6
    nx, ny = npix_shape
    time = synth.make_time(nt)
    cube = synth.make_traces(np.arange(nx * ny) + 11, max(nt, 320))[:, :nt].reshape(nx, ny, nt).copy()
    chain = synth.default_chain(time)
    npix, nf = nx * ny, nt // 2 + 1
    fft = np.zeros((npix, nf, 2), np.float32); amp = np.zeros((npix, nf), np.float32)
    ph = np.zeros((npix, nf), np.float32); out = np.zeros((npix, nt), np.float32); img = np.zeros(npix, np.float32)
    rc = emu.emu_pipeline(nt, C.c_size_t(npix), _p(cube), _p(chain["w_pre"]), _p(chain["fd_mask"]),
                          _p(chain["w_post"]), _p(fft), _p(amp), _p(ph), _p(out), _p(img))
    assert rc == 0
    ref = ob.run_pipeline(cube, time, chain)
    scale = np.abs(ref["fft"]).max()
    assert np.abs(fft.reshape(ref["fft"].shape) - ref["fft"]).max() / scale < 1e-5
    assert np.abs(amp.reshape(ref["amplitudes"].shape) - ref["amplitudes"]).max() / scale < 1e-5
    assert np.abs(out.reshape(ref["data"].shape) - ref["data"]).max() / max(np.abs(ref["data"]).max(), 1e-30) < 1e-5
    assert np.abs(img.reshape(ref["img"].shape) - ref["img"]).max() / max(ref["img"].max(), 1e-30) < 1e-5
    # unwrapped phases on the strong bins of the unmasked spectrum
    st = ob.fft_stage((cube * chain["w_pre"]).astype(np.float32), time, 0, 0.0, 0.0)
    strong = st["amplitudes"] > 0.05 * st["amplitudes"].max(axis=-1, keepdims=True)
    d = ph.reshape(ref["phases"].shape).astype(np.float64) - ref["phases"]
    assert np.abs(d - 2 * np.pi * np.round(d / (2 * np.pi)))[strong].max() < 3e-3




@pytest.mark.parametrize("nt,npix", [(2002, 5), (2400, 3), (3000, 14), (4000, 9), (2002, 1)])
def test_half_length_pipeline_vs_oracle(emu, nt, npix):
    """PH kernels (fft_ph.hpp): an even length whose half is a P plan (2 x 1001, 2 x 1200, 2 x 1500, 2 x 2000) runs as
    a half-length mixed-radix transform + split, one trace per wave, one launch; more traces than a block has waves,
    and the stand-alone forward / inverse kernels land on the fused launch's values"""
    emu.emu_allow_f(1)
    emu.emu_allow_p(1)
    emu.emu_set_p_pairs(1)
    assert emu.emu_half_n(nt) == nt // 2
    time = synth.make_time(nt)
    cube = synth.make_traces(np.arange(npix) + 23, max(nt, 320))[:, :nt].reshape(npix, 1, nt).copy()
    chain = synth.default_chain(time)
    nf = nt // 2 + 1
    fft = np.zeros((npix, nf, 2), np.float32); amp = np.zeros((npix, nf), np.float32)
    ph = np.zeros((npix, nf), np.float32); out = np.zeros((npix, nt), np.float32); img = np.zeros(npix, np.float32)
    assert emu.emu_pipeline(nt, C.c_size_t(npix), _p(cube), _p(chain["w_pre"]), _p(chain["fd_mask"]), _p(chain["w_post"]),
                            _p(fft), _p(amp), _p(ph), _p(out), _p(img)) == 0
    ref = ob.run_pipeline(cube, time, chain)
    scale = np.abs(ref["fft"]).max()
    assert np.abs(fft.reshape(ref["fft"].shape) - ref["fft"]).max() / scale < 1e-5
    assert np.abs(amp.reshape(ref["amplitudes"].shape) - ref["amplitudes"]).max() / scale < 1e-5
    assert np.abs(out.reshape(ref["data"].shape) - ref["data"]).max() / max(np.abs(ref["data"]).max(), 1e-30) < 1e-5
    assert np.abs(img.reshape(ref["img"].shape) - ref["img"]).max() / max(ref["img"].max(), 1e-30) < 1e-5
    assert np.all(fft[:, 0, 1] == 0.0) and np.all(fft[:, -1, 1] == 0.0) and not np.signbit(fft[:, [0, -1], 1]).any()
    st = ob.fft_stage((cube * chain["w_pre"]).astype(np.float32), time, 0, 0.0, 0.0)
    strong = st["amplitudes"] > 0.05 * st["amplitudes"].max(axis=-1, keepdims=True)
    d = ph.reshape(ref["phases"].shape).astype(np.float64) - ref["phases"]
    assert np.abs(d - 2 * np.pi * np.round(d / (2 * np.pi)))[strong].max() < 3e-3
    # the stage entry points: the same spectrum from the forward kernel, the same samples from the inverse kernel
    fft2 = np.zeros_like(fft); amp2 = np.zeros_like(amp); ph2 = np.zeros_like(ph)
    assert emu.emu_fft_fwd(nt, C.c_size_t(npix), _p(cube), _p(chain["w_pre"]), None, None, _p(fft2), _p(amp2), _p(ph2),
                           _p(chain["fd_mask"])) == 0
    assert np.array_equal(fft2, fft) and np.array_equal(amp2, amp) and np.array_equal(ph2, ph)
    out2 = np.zeros_like(out); img2 = np.zeros_like(img)
    assert emu.emu_fft_inv(nt, C.c_size_t(npix), _p(fft), _p(chain["w_post"]), _p(out2), _p(img2)) == 0
    assert np.array_equal(out2, out) and np.array_equal(img2, img)


@pytest.mark.parametrize("nt,npix", [(8200, 3), (8193, 1), (32768, 2)])
def test_long_traces_global_scratch_vs_oracle(emu, nt, npix):
    """trace lengths whose transform buffers do not fit the CU's LDS — not a power of two above 8191, powers of two
    above 16384 — run the G kernels with their buffers in global scratch (round 3: realfft plans any length, these were
    THZ_ERR_UNSUPPORTED); forward, inverse and the two-launch chain against the oracle"""
    emu.emu_allow_f(1)
    emu.emu_allow_p(1)
    emu.emu_set_grid_cap(1)   # one block of four waves: the emulation runs every lane as a host thread
    try:
        time = synth.make_time(nt)
        cube = synth.make_traces(np.arange(npix) + 3, max(nt, 320))[:, :nt].reshape(npix, 1, nt).copy()
        chain = synth.default_chain(time)
        nf = nt // 2 + 1
        fft = np.zeros((npix, nf, 2), np.float32); amp = np.zeros((npix, nf), np.float32)
        ph = np.zeros((npix, nf), np.float32); out = np.zeros((npix, nt), np.float32); img = np.zeros(npix, np.float32)
        assert emu.emu_pipeline(nt, C.c_size_t(npix), _p(cube), _p(chain["w_pre"]), _p(chain["fd_mask"]), _p(chain["w_post"]),
                                _p(fft), _p(amp), _p(ph), _p(out), _p(img)) == 0
    finally:
        emu.emu_set_grid_cap(0)
    ref = ob.run_pipeline(cube, time, chain)
    scale = np.abs(ref["fft"]).max()
    assert np.abs(fft.reshape(ref["fft"].shape) - ref["fft"]).max() / scale < 1e-5
    assert np.abs(amp.reshape(ref["amplitudes"].shape) - ref["amplitudes"]).max() / scale < 1e-5
    assert np.abs(out.reshape(ref["data"].shape) - ref["data"]).max() / max(np.abs(ref["data"]).max(), 1e-30) < 1e-5
    assert np.abs(img.reshape(ref["img"].shape) - ref["img"]).max() / max(ref["img"].max(), 1e-30) < 1e-5
    st = ob.fft_stage((cube * chain["w_pre"]).astype(np.float32), time, 0, 0.0, 0.0)
    strong = st["amplitudes"] > 0.05 * st["amplitudes"].max(axis=-1, keepdims=True)
    d = ph.reshape(ref["phases"].shape).astype(np.float64) - ref["phases"]
    assert np.abs(d - 2 * np.pi * np.round(d / (2 * np.pi)))[strong].max() < 3e-3


@pytest.mark.parametrize("nt", [1024, 4096])
def test_f_kernels_full_window_and_data_out(emu, nt):
    """non-edge windows (Hann: every block != 1) take the F kernels' "full" path;
    asking for the windowed trace splits the multiply into its own launch"""
    emu.emu_allow_f(1)
    npix = 9
    time = synth.make_time(nt)
    x = synth.make_traces(np.arange(npix) + 3, nt)
    w = ob.apply_window(2, np.ones(nt, np.float32), time)
    dout, fft, amp, ph = _fwd(emu, x, w, np.ones(nt // 2 + 1, np.float32), nt)
    st = ob.fft_stage(x.reshape(1, npix, nt), time, 2)
    assert np.array_equal(dout, st["data"][0])
    scale = np.abs(st["fft"]).max()
    assert np.abs(fft - st["fft"][0]).max() / scale < 1e-5
    assert np.abs(amp - st["amplitudes"][0]).max() / scale < 1e-5
    out = np.zeros_like(x)
    img = np.zeros(npix, np.float32)
    assert emu.emu_fft_inv(nt, C.c_size_t(npix), _p(np.ascontiguousarray(st["fft"][0])), _p(w), _p(out), _p(img)) == 0
    back, _ = ob.ifft_stage(st["fft"], nt)
    assert np.abs(out - back[0] * w).max() / np.abs(back).max() < 1e-5


def test_roi_mask_kernel_bit_exact(emu):
    g = np.load(os.path.join(HERE, "golden", "roi_masks.npz"))
    for name in ("concave", "outside_clamped", "convex_cw"):
        poly = (g[name + "_poly"] // np.uint64(1)).astype(np.uint64)
        s0, s1 = 32, 32
        xs, ys = poly[:, 0], poly[:, 1]
        clamp = lambda v, m: min(int(v), m - 1)
        mask = np.zeros((s0, s1), np.uint8)
        emu.emu_roi_mask(_p(np.ascontiguousarray(poly)), poly.shape[0],
                         C.c_uint64(clamp(xs.min(), s1)), C.c_uint64(clamp(xs.max(), s1)),
                         C.c_uint64(clamp(ys.min(), s0)), C.c_uint64(clamp(ys.max(), s0)),
                         C.c_uint64(s1), C.c_uint64(s0), _p(mask))
        assert np.array_equal(mask, g[f"{name}_32x32_s1_mask"]), name


def test_fast_atan2_accuracy(emu):
    """the epilogue's lean atan2 against numpy float64: <= 3.2e-7 rad everywhere,
    correct quadrants and axis values"""
    rng = np.random.default_rng(0)
    n = 400000
    x = (rng.standard_normal(n) * 10 ** rng.uniform(-6, 6, n)).astype(np.float32)
    y = (rng.standard_normal(n) * 10 ** rng.uniform(-6, 6, n)).astype(np.float32)
    x[:8] = [1, -1, 0, 0, 1, -1, 0, 2]
    y[:8] = [0, 0, 1, -1, 1, 1, 0, -2]
    out = np.empty(n, np.float32)
    emu.emu_fast_atan2(_p(y), _p(x), n, _p(out))
    ref = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    assert np.abs(out - ref).max() < 3.2e-7  # half an ulp of pi is 1.2e-7 of it
    assert out[6] == 0.0 and abs(out[1] - np.pi) < 3e-7 and abs(out[3] + np.pi / 2) < 3e-7


def test_div_const_edge_cases(emu):
    """DivConst (thz_device.hpp: x / d in three instructions for a loop-invariant d): the correctly rounded
    quotient for every finite operand with a normal quotient — and, documented there as outside its contract,
    NaN for an infinite operand (IEEE: the infinity) and at most one ulp off in the denormal range"""
    rng = np.random.default_rng(5)
    for d in (1001.0, 777.0, 2000.0, 8191.0, 3.0):
        x = (rng.standard_normal(200000) * 10 ** rng.uniform(-20, 20, 200000)).astype(np.float32)
        out = np.empty_like(x)
        emu.emu_div_const(_p(x), x.size, C.c_float(d), _p(out))
        assert np.array_equal(out, x / np.float32(d))
        edge = np.array([np.inf, -np.inf, 0.0, -0.0, np.nan], np.float32)
        oe = np.empty_like(edge)
        emu.emu_div_const(_p(edge), edge.size, C.c_float(d), _p(oe))
        assert np.isnan(oe[0]) and np.isnan(oe[1]) and np.isnan(oe[4])          # the documented non-IEEE cases
        assert oe[2] == 0.0 and oe[3] == 0.0                                      # (-0 comes back as +0: same value)
        tiny = (rng.standard_normal(20000) * 1e-38).astype(np.float32)               # quotients in the denormal range
        ot = np.empty_like(tiny)
        emu.emu_div_const(_p(tiny), tiny.size, C.c_float(d), _p(ot))
        ulp = np.float32(1.4e-45)
        assert np.abs(ot.astype(np.float64) - (tiny / np.float32(d)).astype(np.float64)).max() <= float(ulp) * 1.01


def _rl_reference(d, u, psf, mode):
    """one Richardson-Lucy iteration with the reference's loops (deconvolution.rs:432-458 for kernels of at
    most 256 elements: correlation-indexed, m outer / n inner, f32, no FMA; the 'same' true convolution the
    FFT branch stands for otherwise), numpy float32 scalars so that every rounding is the reference's"""
    H, W = d.shape
    pr, pc = psf.shape
    f32 = np.float32

    def conv(a, k):
        out = np.zeros((H, W), f32)
        for i in range(H):
            for j in range(W):
                s = f32(0)
                for m in range(pr):
                    x = i + m - pr // 2 if mode == 0 else i + (pr - 1) // 2 - m
                    if x < 0 or x >= H:
                        continue
                    for n in range(pc):
                        y = j + n - pc // 2 if mode == 0 else j + (pc - 1) // 2 - n
                        if 0 <= y < W:
                            s = f32(s + f32(a[x, y] * k[m, n]))
                out[i, j] = s
        return out

    t = (d / (conv(u, psf) + f32(1e-12))).astype(f32)
    return t, (u * conv(t, psf[::-1, ::-1])).astype(f32)


@pytest.mark.parametrize("case", [dict(h=14, w=19, pr=7, pc=9, mode=0), dict(h=9, w=21, pr=13, pc=17, mode=0),
                                  dict(h=12, w=12, pr=3, pc=5, mode=0), dict(h=11, w=13, pr=19, pc=21, mode=1)])
def test_rl_step_kernels(emu, case):
    """k_rl_step (every tap from memory) and k_rl_step_tiled (LDS tile, chunked taps, end-aligned last chunk;
    for wide kernels four pixels per thread and a 16-way row split): narrow kernels bit for bit in the
    reference's order, wide ones within rounding"""
    h, w, pr, pc, mode = (case[k] for k in ("h", "w", "pr", "pc", "mode"))
    assert (pr * pc > 256) == (mode == 1)
    rng = np.random.default_rng(pr * 100 + pc)
    H, W = h + 2 * (pr // 2), w + 2 * (pc // 2)
    d = (0.5 + rng.random((H, W))).astype(np.float32)
    u = (0.5 + rng.random((H, W))).astype(np.float32)
    psf = rng.random((pr, pc)).astype(np.float32)
    psf /= psf.max()
    res = []
    for tiled in (0, 1):
        t, un = np.empty((H, W), np.float32), np.empty((H, W), np.float32)
        assert emu.emu_rl_iteration(h, w, pr, pc, mode, _p(psf), _p(d), _p(u), tiled, _p(t), _p(un)) == 0
        res.append((t, un))
    rt, ru = _rl_reference(d, u, psf, mode)
    if mode == 0:
        for t, un in res:
            assert np.array_equal(t, rt) and np.array_equal(un, ru)
    else:
        assert np.array_equal(res[0][0], rt) and np.array_equal(res[0][1], ru)   # untiled: the same loops
        assert np.abs(res[1][0] - rt).max() / np.abs(rt).max() < 1e-5
        assert np.abs(res[1][1] - ru).max() / np.abs(ru).max() < 1e-5


@pytest.mark.parametrize("case", [dict(h=11, w=13, pr=19, pc=21), dict(h=20, w=37, pr=47, pc=57),
                                  dict(h=9, w=40, pr=17, pc=33), dict(h=40, w=9, pr=53, pc=5),
                                  dict(h=6, w=70, pr=3, pc=121)])   # halo wider than two column passes of a wave
def test_rl_step_separable_kernel(emu, case):
    """k_rl_step_sep: a wide kernel that is an outer product fx x fy (every band PSF of the reference is one)
    as a pass along the rows and a pass down the columns — against the reference's loops over the 2-D array and
    against the 2-D tiled kernel, within rounding; tiles at the ragged edges, halo rows not a multiple of 16,
    a profile that fills its last chunk exactly (pc = 33 -> 48 staged taps) and one shorter than a chunk"""
    h, w, pr, pc = (case[k] for k in ("h", "w", "pr", "pc"))
    assert pr * pc > 256
    rng = np.random.default_rng(pr * 1000 + pc)
    H, W = h + 2 * (pr // 2), w + 2 * (pc // 2)
    d = (0.5 + rng.random((H, W))).astype(np.float32)
    u = (0.5 + rng.random((H, W))).astype(np.float32)
    x, y = np.arange(pr) - pr // 2, np.arange(pc) - pc // 2
    fx = np.exp(-(x - 0.7) ** 2 / (2 * (pr / 5) ** 2)).astype(np.float32)   # off-centre: the mirror differs
    fy = np.exp(-(y + 1.3) ** 2 / (2 * (pc / 6) ** 2)).astype(np.float32)
    psf = np.outer(fx, fy).astype(np.float32)
    res = {}
    for tiled in (0, 1, 2):
        t, un = np.empty((H, W), np.float32), np.empty((H, W), np.float32)
        assert emu.emu_rl_iteration_sep(h, w, pr, pc, 1, _p(psf), _p(fx), _p(fy), _p(d), _p(u), tiled, _p(t), _p(un)) == 0
        res[tiled] = (t, un)
    a64 = lambda a, k: _conv_same_f64(a, k)
    t64 = d / (a64(u, psf) + 1e-12)
    u64 = u * a64(t64, psf[::-1, ::-1])
    for tiled in (0, 1, 2):
        et = np.abs(res[tiled][0] - t64).max() / np.abs(t64).max()
        eu = np.abs(res[tiled][1] - u64).max() / np.abs(u64).max()
        assert et < 2e-6 and eu < 4e-6, (tiled, et, eu)
    # the two passes round no worse than the 2-D sums do
    e2 = np.abs(res[1][1] - u64).max()
    es = np.abs(res[2][1] - u64).max()
    assert es <= 2 * e2 + 1e-7 * np.abs(u64).max()


@pytest.mark.parametrize("case", [dict(h=14, w=19, pr=7, pc=9), dict(h=9, w=41, pr=13, pc=17), dict(h=35, w=12, pr=3, pc=5),
                                  dict(h=33, w=34, pr=15, pc=17)])
def test_rl_step_separable_kernel_narrow_mode(emu, case):
    """k_rl_step_sep in mode 0: a kernel of <= 256 taps — in the reference a direct sum, correlation-indexed
    (deconvolution.rs:432-458) — that is an outer product, as the same two 1-D passes with the profiles the other way
    round; against the reference-order kernels (which are the reference's loops bit for bit) within rounding, with
    off-centre profiles so that a wrong direction in either step or either axis would show"""
    h, w, pr, pc = (case[k] for k in ("h", "w", "pr", "pc"))
    assert pr * pc <= 256 and pr % 2 == 1 and pc % 2 == 1
    rng = np.random.default_rng(pr * 1000 + pc)
    H, W = h + 2 * (pr // 2), w + 2 * (pc // 2)
    d = (0.5 + rng.random((H, W))).astype(np.float32)
    u = (0.5 + rng.random((H, W))).astype(np.float32)
    x, y = np.arange(pr) - pr // 2, np.arange(pc) - pc // 2
    fx = np.exp(-(x - 0.7) ** 2 / (2 * (pr / 5) ** 2)).astype(np.float32)
    fy = np.exp(-(y + 1.3) ** 2 / (2 * (pc / 6) ** 2)).astype(np.float32)
    psf = np.outer(fx, fy).astype(np.float32)
    res = {}
    for tiled in (1, 2):   # 1: k_rl_step_tiled<false> (reference order), 2: k_rl_step_sep
        t, un = np.empty((H, W), np.float32), np.empty((H, W), np.float32)
        assert emu.emu_rl_iteration_sep(h, w, pr, pc, 0, _p(psf), _p(fx), _p(fy), _p(d), _p(u), tiled, _p(t), _p(un)) == 0
        res[tiled] = (t, un)
    rt, ru = _rl_reference(d, u, psf, 0)
    assert np.array_equal(res[1][0], rt) and np.array_equal(res[1][1], ru)
    assert np.abs(res[2][0] - rt).max() / np.abs(rt).max() < 2e-6
    assert np.abs(res[2][1] - ru).max() / np.abs(ru).max() < 4e-6


def _conv_same_f64(a, k):
    """'same' linear convolution in float64: rows / columns [(b-1)/2, (b-1)/2 + a) of the full one"""
    from scipy.signal import convolve2d
    full = convolve2d(a.astype(np.float64), k.astype(np.float64), mode="full")
    sr, sc = (k.shape[0] - 1) // 2, (k.shape[1] - 1) // 2
    return full[sr:sr + a.shape[0], sc:sc + a.shape[1]]


@pytest.mark.parametrize("M,nt", [(1024, 300), (2048, 1001), (4096, 2000)])
def test_deconvolution_transform_kernels(emu, M, nt):
    """k_dc_fft -> k_dc_energy(_f) -> k_dc_combine(_f): band energies of FIR-filtered traces over the 'same'
    slice and the gain-weighted recombination, register-resident core against the generic LDS transform and
    both against numpy float64"""
    rng = np.random.default_rng(M + nt)
    npix, nb, taps = 5, 3, 499
    shift = (taps - 1) // 2
    x = synth.make_traces(np.arange(npix) + 3, max(nt, 320))[:, :nt].astype(np.float32)
    h = (rng.standard_normal((nb, taps)) * np.hanning(taps)).astype(np.float32) / 20
    Hs = (np.fft.rfft(h.astype(np.float64), M, axis=-1) / M)
    H = np.stack([Hs.real, Hs.imag], -1).astype(np.float32)
    gain = (0.5 + rng.random((nb, npix))).astype(np.float32)
    y = np.stack([[np.convolve(x[p].astype(np.float64), h[b].astype(np.float64))[shift:shift + nt] for p in range(npix)]
                  for b in range(nb)])                                   # (nb, npix, nt) 'same' slices
    e_ref = (y ** 2).sum(-1)
    o_ref = (gain[..., None] * y).sum(0)
    res = []
    for use_f in (0, 1):
        en = np.zeros((nb, npix), np.float32); out = np.zeros((npix, nt), np.float32); img = np.zeros(npix, np.float32)
        rc = emu.emu_dc_chain(M, nt, C.c_size_t(npix), nb, shift, _p(x), _p(H), _p(gain), use_f, _p(en), _p(out), _p(img))
        assert rc == 0
        assert np.abs(en - e_ref).max() / e_ref.max() < 2e-5
        assert np.abs(out - o_ref).max() / np.abs(o_ref).max() < 2e-5
        assert np.abs(img - (o_ref ** 2).sum(-1)).max() / (o_ref ** 2).sum(-1).max() < 2e-5
        res.append((en, out))
    assert np.abs(res[0][0] - res[1][0]).max() / e_ref.max() < 5e-6


@pytest.mark.parametrize("M,nt,kind", [(1024, 300, "pulse"), (2048, 1001, "pulse"), (2048, 1001, "noise"),
                                       (2048, 1001, "edges"), (4096, 2000, "pulse"), (1024, 100, "noise"),
                                       (8192, 4000, "pulse"), (16384, 8000, "noise"), (2048, 1001, "middle")])
def test_deconvolution_band_energies_parseval_form(emu, M, nt, kind):
    """k_dc_energy_pv: band energies over the 'same' slice as Parseval's sum over |X|^2 c_k M |H_b|^2 minus the
    energies of the first / last 249 samples of the full convolution (one 512-point complex transform per band),
    against numpy float64 and against the transform-per-band kernel; pulses, noise up to both ends of the trace
    (head and tail corrections of the size of the result) and a trace that lives only in its first and last samples"""
    rng = np.random.default_rng(M + nt)
    npix, nb, taps = 6, 4, 499
    shift = (taps - 1) // 2
    if kind == "pulse":
        x = synth.make_traces(np.arange(npix) + 3, max(nt, 320))[:, :nt].astype(np.float32)
    elif kind == "noise":
        x = rng.standard_normal((npix, nt)).astype(np.float32)
    elif kind == "middle":   # nothing in the first / last 249 samples: no edges to subtract (the kernel skips their transforms)
        x = rng.standard_normal((npix, nt)).astype(np.float32)
        x[:, :260] = 0.0
        x[:, -255:] = 0.0
        x[0, 250] = 0.0; x[1, :] = 0.0   # ... and a trace that is zero altogether
    else:
        x = np.zeros((npix, nt), np.float32)
        x[:, :5] = rng.standard_normal((npix, 5))
        x[:, -7:] = rng.standard_normal((npix, 7))
    h = (rng.standard_normal((nb, taps)) * np.hanning(taps)).astype(np.float32) / 20
    h = (0.5 * (h + h[:, ::-1])).astype(np.float32)      # linear phase, as the Kaiser bank's filters are
    h[-1] = (rng.standard_normal(taps) / 20).astype(np.float32)   # ... and one that is not: the form does not need it
    y = np.stack([[np.convolve(x[p].astype(np.float64), h[b].astype(np.float64))[shift:shift + nt] for p in range(npix)]
                  for b in range(nb)])
    e_ref = (y ** 2).sum(-1)
    en = np.zeros((nb, npix), np.float32)
    rc = emu.emu_dc_energy_pv(M, nt, C.c_size_t(npix), nb, taps, _p(x), _p(h), _p(en))
    assert rc == 0
    if kind in ("noise", "middle") and M == 2048:
        # one block for all pixels: several batches per block, the table's halves alternating across them
        npix2, nb2 = 19, 3
        x2 = np.concatenate([x, rng.standard_normal((npix2 - npix, nt)).astype(np.float32)])
        e2_ref = np.stack([[(np.convolve(x2[p].astype(np.float64), h[b].astype(np.float64))[shift:shift + nt] ** 2).sum()
                            for p in range(npix2)] for b in range(nb2)])
        en2 = np.zeros((nb2, npix2), np.float32)
        emu.emu_set_grid_cap(1)
        try:
            assert emu.emu_dc_energy_pv(M, nt, C.c_size_t(npix2), nb2, taps, _p(x2), _p(h[:nb2].copy()), _p(en2)) == 0
        finally:
            emu.emu_set_grid_cap(0)
        assert np.abs(en2 - e2_ref).max() / e2_ref.max() < 5e-6
    err = np.abs(en - e_ref).max() / e_ref.max()
    assert err < 5e-6, err
    if kind == "middle":
        assert np.all(en[:, 1] == 0.0)
    # against the kernel it replaces (same forward transform, same f32 filter spectra)
    Hs = np.fft.rfft(h.astype(np.float64), M, axis=-1) / M
    H = np.stack([Hs.real, Hs.imag], -1).astype(np.float32)
    gain = np.ones((nb, npix), np.float32)
    en0 = np.zeros((nb, npix), np.float32); out = np.zeros((npix, nt), np.float32); img = np.zeros(npix, np.float32)
    use_f = 1 if M <= 4096 else 0   # padded lengths above 4096 have no F core: the generic LDS transform
    assert emu.emu_dc_chain(M, nt, C.c_size_t(npix), nb, shift, _p(x), _p(H), _p(gain), use_f, _p(en0), _p(out), _p(img)) == 0
    assert np.abs(en - en0).max() / e_ref.max() < 5e-6


def test_helper_kernels(emu):
    """the bandwidth-shaped kernels around the transforms, each in its batched / wave-per-trace form:
    window multiply, two-level column sums (ragged row length), pixel-list sums in list order, block means
    with the ragged-edge rule, tilt copy — against numpy / the oracle, bit for bit where the order is fixed"""
    rng = np.random.default_rng(5)
    # K6 multiplier: odd length, more traces than one block handles
    npix, nt = 37, 1001
    x = rng.standard_normal((npix, nt)).astype(np.float32)
    wv = rng.random(nt).astype(np.float32)
    out = np.empty_like(x)
    assert emu.emu_td_window(C.c_size_t(npix), nt, _p(x), _p(wv), _p(out)) == 0
    assert np.array_equal(out, x * wv)
    for nt2 in (256, 1024, 4096):   # whole rounds of a wave's 16-byte accesses: the window chunks live in registers
        x2 = rng.standard_normal((npix, nt2)).astype(np.float32)
        w2 = rng.random(nt2).astype(np.float32)
        o2 = np.empty_like(x2)
        assert emu.emu_td_window(C.c_size_t(npix), nt2, _p(x2), _p(w2), _p(o2)) == 0
        assert np.array_equal(o2, x2 * w2)
    # K8 sums: rows of nf and 2 nf floats (only 4-byte aligned), above and below the two-level threshold
    for nrows, L in ((300, 513), (70, 1026), (40, 129), (9000, 130)):
        a = rng.standard_normal((nrows, L)).astype(np.float32)
        s = np.empty(L, np.float32)
        assert emu.emu_pixel_sum(C.c_size_t(nrows), C.c_size_t(L), _p(a), _p(s)) == 0
        ref = a.astype(np.float64).sum(0)
        assert np.abs(s - ref).max() <= 2e-6 * np.abs(a).sum(0).max()
    # K9 pixel-list mean: the adds in list order (the reference's iteration order)
    n_pix, ln = 500, 257
    a = rng.standard_normal((n_pix, ln)).astype(np.float32)
    for count in (1, 63, 64, 65, 200):
        lst = rng.permutation(n_pix)[:count].astype(np.uint32)
        s = np.empty(ln, np.float32)
        assert emu.emu_gather_sum(_p(a), C.c_size_t(ln), _p(lst), count, C.c_float(float(count)), _p(s)) == 0
        acc = np.zeros(ln, np.float32)
        for i in lst:
            acc = acc + a[i]
        assert np.array_equal(s, acc / np.float32(count))
    # K10 block means incl. ragged edges (sum / s^2 whatever the block holds)
    nx, ny, L, sf = 7, 9, 130, 2
    a = rng.standard_normal((nx, ny, L)).astype(np.float32)
    o = np.empty((nx // sf, ny // sf, L), np.float32)
    assert emu.emu_scale3d(_p(a), C.c_size_t(nx), C.c_size_t(ny), C.c_size_t(L), C.c_size_t(sf), _p(o)) == 0
    assert np.array_equal(o, ob.scale3d(a, sf))
    for L2, sf2 in ((132, 2), (1024, 3)):   # rows of whole 16-byte chunks: the vector path; s = 3: a divisor that is no power of two
        a2 = rng.standard_normal((nx, ny, L2)).astype(np.float32)
        o2 = np.empty((nx // sf2, ny // sf2, L2), np.float32)
        assert emu.emu_scale3d(_p(a2), C.c_size_t(nx), C.c_size_t(ny), C.c_size_t(L2), C.c_size_t(sf2), _p(o2)) == 0
        assert np.array_equal(o2, ob.scale3d(a2, sf2))
    # K11 tilt copy: front fill with the first sample, tapered trace at its insert index, clipped at the end
    npix, nt_in, nt_out = 13, 300, 340
    x = rng.standard_normal((npix, nt_in)).astype(np.float32)
    taper = rng.random(nt_in).astype(np.float32)
    ins = rng.integers(0, 60, npix).astype(np.int32)
    o = np.empty((npix, nt_out), np.float32)
    assert emu.emu_tilt(C.c_size_t(npix), nt_in, nt_out, _p(x), _p(taper), _p(ins), _p(o)) == 0
    ref = np.zeros((npix, nt_out), np.float32)
    for p in range(npix):
        ref[p, :ins[p]] = x[p, 0]
        n = min(nt_in, nt_out - ins[p])
        ref[p, ins[p]:ins[p] + n] = (x[p] * taper)[:n]
    assert np.array_equal(o, ref)


def test_filter_spectra_kernel(emu):
    """H_b[k] = (1/M) sum_j h_b[j] exp(-2 pi i j k / M): the kernel's table-stepped double sums vs numpy's rfft"""
    rng = np.random.default_rng(11)
    nb, taps, M = 3, 499, 2048
    h = rng.standard_normal((nb, taps)).astype(np.float32)
    H = np.zeros((nb, M // 2 + 1, 2), np.float32)
    assert emu.emu_dc_filter_spectra(_p(h), nb, taps, M, _p(H)) == 0
    ref = np.fft.rfft(h.astype(np.float64), M, axis=-1) / M
    assert np.abs((H[..., 0] + 1j * H[..., 1]) - ref).max() / np.abs(ref).max() < 2e-7


def _wiener_cmask(time, nf):
    """K13 multiplier from the synthetic reference pulse (the noise-free template of SURVEY §8d), in numpy fp64
    -> f32: H = conj(R) / (|R|^2 + eps max|R|^2), R = rfft(window * reference)"""
    z = ((time - time[0] - 11.0) / 0.35).astype(np.float64)
    ref = -z * np.exp(-z * z)
    w = ob.apply_window(0, np.ones(time.size, np.float32), time, 1.0, 7.0).astype(np.float64)
    R = np.fft.rfft(ref * w)
    H = np.conj(R) / (np.abs(R) ** 2 + 1e-2 * (np.abs(R) ** 2).max())
    out = np.empty((nf, 2), np.float32)
    out[:, 0] = H.real
    out[:, 1] = H.imag
    return out


@pytest.mark.parametrize("bar", [0, 1, 2, 3])
@pytest.mark.parametrize("mode", ["plain", "cmask"])
@pytest.mark.parametrize("nt", [1024, 2048, 4096])
def test_fused_pipeline_cmask_barriers(emu, nt, mode, bar):
    """k_f<pipe> with the complex per-bin multiplier (K13) and the store-phase barriers: same results as the
    plain fused chain / a numpy fp64 model of it"""
    emu.emu_allow_f(1)
    emu.emu_set_f_bar(bar)
    try:
        nx, ny = 3, 7   # 21 traces: ragged last round in every block shape (8, 7 or 6 waves)
        time, cube = synth.make_cube(nx, ny, nt)
        chain = synth.default_chain(time, backend=None)
        npix, nf = nx * ny, nt // 2 + 1
        H = _wiener_cmask(time, nf) if "cmask" in mode else None
        fft = np.zeros((npix, nf, 2), np.float32); amp = np.zeros((npix, nf), np.float32)
        ph = np.zeros((npix, nf), np.float32); out = np.zeros((npix, nt), np.float32); img = np.zeros(npix, np.float32)
        rc = emu.emu_pipeline_ex(nt, C.c_size_t(npix), _p(cube), _p(chain["w_pre"]), _p(chain["fd_mask"]), _p(H),
                                 _p(chain["w_post"]), _p(fft), _p(amp), _p(ph), _p(out), _p(img))
        assert rc == 0
        ref = ob.run_pipeline(cube, time, chain)
        scale = np.abs(ref["fft"]).max()
        if H is None:
            assert np.abs(fft.reshape(ref["fft"].shape) - ref["fft"]).max() / scale < 1e-5
            assert np.abs(amp.reshape(ref["amplitudes"].shape) - ref["amplitudes"]).max() / scale < 1e-5
            assert np.abs(out.reshape(ref["data"].shape) - ref["data"]).max() / np.abs(ref["data"]).max() < 1e-5
        else:
            # fp64 model: Y = rfft(pre * x) * H * mask with real DC / Nyquist, irfft, post window
            X = np.fft.rfft(cube.reshape(npix, nt).astype(np.float64) * chain["w_pre"].astype(np.float64), axis=1)
            Hc = (H[:, 0].astype(np.float64) + 1j * H[:, 1]) * chain["fd_mask"]
            Y = X * Hc
            a_ref = np.abs(Y)
            Y[:, 0] = Y[:, 0].real
            Y[:, -1] = Y[:, -1].real
            t_ref = np.fft.irfft(Y, n=nt, axis=1) * chain["w_post"]
            got = fft[..., 0] + 1j * fft[..., 1]
            assert np.abs(got - Y).max() / np.abs(Y).max() < 1e-5
            assert np.all(fft[:, 0, 1] == 0) and np.all(fft[:, -1, 1] == 0)
            assert np.abs(amp - a_ref).max() / a_ref.max() < 1e-5
            assert np.abs(out - t_ref).max() / np.abs(t_ref).max() < 1e-5
            assert np.abs(img - (t_ref ** 2).sum(1)).max() / (t_ref ** 2).sum(1).max() < 1e-5
        # phases are those of X in every mode
        d = ph.reshape(ref["phases"].shape) - ref["phases"]
        strong = ref["amplitudes"] > 0.05 * ref["amplitudes"].max(axis=-1, keepdims=True)
        d = d - 2 * np.pi * np.round(d / (2 * np.pi))
        assert np.abs(d[strong | (np.abs(d) < 1)]).max() < 3e-3
    finally:
        emu.emu_set_f_bar(-1)


@pytest.mark.parametrize("nt", [1024, 4096])
def test_forward_inverse_with_store_barriers(emu, nt):
    """k_f<fwd> / k_f<inv> with kCfgBar: identical outputs to the barrier-free kernels (the barrier only aligns
    the block's store phases), incl. a ragged last round where some waves take the barriers without a trace"""
    emu.emu_allow_f(1)
    npix = 13
    time = synth.make_time(nt)
    x = np.ascontiguousarray(synth.make_traces(np.arange(npix) + 3, nt), np.float32)
    wa = ob.apply_window(0, np.ones(nt, np.float32), time, 1.0, 7.0)
    mask = ob.fd_bandpass_window(ob.frequency_axis(time), 0.2, 5.0, 0.1)[0]
    win = ob.td_bandpass_window(time, float(time[0]), float(time[-1]), 0.1)[0]
    res = {}
    try:
        for bar in (0, 3):
            emu.emu_set_f_bar(bar)
            fft = np.zeros((npix, nt // 2 + 1, 2), np.float32); amp = np.zeros((npix, nt // 2 + 1), np.float32)
            ph = np.zeros_like(amp)
            assert emu.emu_fft_fwd(nt, C.c_size_t(npix), _p(x), _p(wa), None, None, _p(fft), _p(amp), _p(ph), _p(mask)) == 0
            out = np.zeros_like(x); img = np.zeros(npix, np.float32)
            assert emu.emu_fft_inv(nt, C.c_size_t(npix), _p(fft), _p(win), _p(out), _p(img)) == 0
            res[bar] = (fft, amp, ph, out, img)
    finally:
        emu.emu_set_f_bar(-1)
    for a, b in zip(res[0], res[3]):
        assert np.array_equal(a, b)
    assert np.abs(res[0][3]).max() > 0


@pytest.mark.parametrize("bar", [0, 3])
@pytest.mark.parametrize("mode", ["plain", "cmask"])
@pytest.mark.parametrize("nt,npix,cap", [(1024, 21, 0), (2048, 9, 0), (4096, 17, 0), (4096, 3, 0), (4096, 17, 1), (1024, 37, 2)])
def test_fused_pipeline_in_kernel_pixel_sums(emu, nt, npix, cap, mode, bar):
    """k_f<pipe, kCfgSums>: the block's waves add their amplitudes and unwrapped phases to ONE set of accumulators in
    LDS, group by group in ticket order (FSums).  Every output equals the plain fused chain's bit for bit, and the
    sums equal the column sums of the stored amplitude / phase arrays (another order of f32 additions: 2e-6).
    Trace counts leave ragged last rounds (the waves without a trace stay away) and, for 3 traces, waves that never
    see a trace; cap: at most that many blocks, so that a block's waves go through several rounds (tickets of
    successive rounds, the last round ragged)."""
    emu.emu_allow_f(1)
    emu.emu_set_f_bar(bar)
    emu.emu_set_grid_cap(cap)
    try:
        time = synth.make_time(nt)
        cube = synth.make_traces(np.arange(npix) + 17, nt).reshape(npix, 1, nt).copy()
        chain = synth.default_chain(time, backend=None)
        nf = nt // 2 + 1
        H = _wiener_cmask(time, nf) if mode == "cmask" else None
        outs = []
        for with_sums in (False, True):
            fft = np.zeros((npix, nf, 2), np.float32); amp = np.zeros((npix, nf), np.float32)
            ph = np.zeros((npix, nf), np.float32); out = np.zeros((npix, nt), np.float32); img = np.zeros(npix, np.float32)
            sums = np.full(2 * nf, np.nan, np.float32)
            if with_sums:
                rows = emu.emu_pipeline_sums(nt, C.c_size_t(npix), _p(cube), _p(chain["w_pre"]), _p(chain["fd_mask"]), _p(H),
                                             _p(chain["w_post"]), _p(fft), _p(amp), _p(ph), _p(out), _p(img), _p(sums))
                assert rows >= 1
            else:
                assert emu.emu_pipeline_ex(nt, C.c_size_t(npix), _p(cube), _p(chain["w_pre"]), _p(chain["fd_mask"]), _p(H),
                                           _p(chain["w_post"]), _p(fft), _p(amp), _p(ph), _p(out), _p(img)) == 0
            outs.append((fft, amp, ph, out, img, sums))
        for a, b in zip(outs[0][:5], outs[1][:5]):
            assert np.array_equal(a, b)
        amp, ph, sums = outs[1][1], outs[1][2], outs[1][5]
        assert np.isfinite(sums).all()
        sa, sp = amp.astype(np.float64).sum(0), ph.astype(np.float64).sum(0)
        assert np.abs(sums[:nf] - sa).max() <= 2e-6 * np.abs(sa).max()
        assert np.abs(sums[nf:] - sp).max() <= 2e-6 * np.abs(sp).max()
        assert np.abs(sa).max() > 0 and np.abs(sp).max() > 0
    finally:
        emu.emu_set_f_bar(-1)
        emu.emu_set_grid_cap(0)


@pytest.mark.parametrize("band", [(0.2, 5.0), (0.0, 4.9), (2.0, 2.3), (4.0, 9.0)])
def test_fused_pipeline_band_limited_complex_multiplier(emu, band):
    """kCfgBand (round 3): at nt = 4096 the chain with a complex multiplier AND the in-launch sums stages only the bins
    where the real band pass is not zero — between two quads of zeros every other bin's index is clamped to — and so
    keeps its eighth wave.  Same products, same outputs, bit for bit, as the full table; band edges that are not
    multiples of four, a band that starts at bin 0, one that reaches the last bin (too wide for the table: the launcher
    falls back to the full one)"""
    emu.emu_allow_f(1); emu.emu_allow_p(1)
    nt, npix = 4096, 11
    nf = nt // 2 + 1
    time = synth.make_time(nt)
    cube = synth.make_traces(np.arange(npix) + 3, nt)
    chain = synth.oracle_chain(time)
    freq = ob.frequency_axis(time)
    mask, lo, hi = ob.fd_bandpass_window(freq, band[0], band[1], 0.1)
    rng = np.random.default_rng(4)
    H = rng.standard_normal((nf, 2)).astype(np.float32)
    outs = []
    for banded in (False, True):
        fft = np.zeros((npix, nf, 2), np.float32); amp = np.zeros((npix, nf), np.float32)
        ph = np.zeros((npix, nf), np.float32); out = np.zeros((npix, nt), np.float32); img = np.zeros(npix, np.float32)
        sums = np.full(2 * nf, np.nan, np.float32)
        if banded:
            rows = emu.emu_pipeline_sums_band(nt, C.c_size_t(npix), _p(cube), _p(chain["w_pre"]), _p(mask), _p(H), _p(chain["w_post"]),
                                              _p(fft), _p(amp), _p(ph), _p(out), _p(img), _p(sums), int(lo), int(hi))
        else:
            rows = emu.emu_pipeline_sums(nt, C.c_size_t(npix), _p(cube), _p(chain["w_pre"]), _p(mask), _p(H), _p(chain["w_post"]),
                                         _p(fft), _p(amp), _p(ph), _p(out), _p(img), _p(sums))
        assert rows >= 1
        outs.append((fft, amp, ph, out, img, sums))
    for a, b in zip(outs[0][:5], outs[1][:5]):
        assert np.array_equal(a, b)
    # the sums: seven waves per block against eight add the traces in another order
    sa, sb = outs[0][5].astype(np.float64), outs[1][5]
    assert np.abs(sa - sb).max() <= 2e-6 * np.abs(sa).max()
    assert np.abs(outs[1][0]).max() > 0


@pytest.mark.parametrize("mode", ["plain", "cmask"])
@pytest.mark.parametrize("nt,npix,cap", [(1001, 37, 0), (1000, 40, 0), (1001, 2, 0), (1001, 1, 0), (1001, 75, 1)])
def test_mixed_radix_pipeline_in_kernel_pixel_sums(emu, nt, npix, cap, mode):
    """k_p<pipe, SUMS> (PSums): the pixel sums of the stored amplitudes and unwrapped phases taken inside the launch
    by ticket-ordered accumulation in LDS — outputs bit-identical to the launch without them, sums equal to the
    column sums of the stored arrays; odd trace counts (a pair with one trace), more pairs than one block has waves
    (cap = 1: one block, three trips, the last one ragged), fewer than two waves' worth"""
    emu.emu_allow_f(1); emu.emu_allow_p(1)
    emu.emu_set_p_pairs(1)
    emu.emu_set_grid_cap(cap)
    try:
        assert emu.emu_family(nt) == 6
        time = synth.make_time(nt)
        cube = synth.make_traces(np.arange(npix) + 23, max(nt, 1024))[:, :nt].reshape(npix, 1, nt).copy()
        chain = synth.default_chain(time)
        nf = nt // 2 + 1
        H = _wiener_cmask(time, nf) if mode == "cmask" else None
        outs = []
        for with_sums in (False, True):
            fft = np.zeros((npix, nf, 2), np.float32); amp = np.zeros((npix, nf), np.float32)
            ph = np.zeros((npix, nf), np.float32); out = np.zeros((npix, nt), np.float32); img = np.zeros(npix, np.float32)
            sums = np.full(2 * nf, np.nan, np.float32)
            if with_sums:
                assert emu.emu_pipeline_sums(nt, C.c_size_t(npix), _p(cube), _p(chain["w_pre"]), _p(chain["fd_mask"]), _p(H),
                                             _p(chain["w_post"]), _p(fft), _p(amp), _p(ph), _p(out), _p(img), _p(sums)) >= 1
            else:
                assert emu.emu_pipeline_ex(nt, C.c_size_t(npix), _p(cube), _p(chain["w_pre"]), _p(chain["fd_mask"]), _p(H),
                                           _p(chain["w_post"]), _p(fft), _p(amp), _p(ph), _p(out), _p(img)) == 0
            outs.append((fft, amp, ph, out, img, sums))
        for a, b in zip(outs[0][:5], outs[1][:5]):
            assert np.array_equal(a, b)
        amp, ph, sums = outs[1][1], outs[1][2], outs[1][5]
        assert np.isfinite(sums).all()
        sa, sp = amp.astype(np.float64).sum(0), ph.astype(np.float64).sum(0)
        assert np.abs(sums[:nf] - sa).max() <= 2e-6 * np.abs(sa).max()
        assert np.abs(sums[nf:] - sp).max() <= 2e-6 * max(np.abs(sp).max(), 1.0)
        assert np.abs(sa).max() > 0
    finally:
        emu.emu_set_p_pairs(0)
        emu.emu_set_grid_cap(0)


@pytest.mark.parametrize("pairs", [1, 2])
@pytest.mark.parametrize("nt", [1001, 1000])
def test_mixed_radix_pipeline_complex_multiplier(emu, nt, pairs):
    """k_p<pipe, CM> — the complex per-bin multiplier (K13, reference deconvolution) inside the mixed-radix chain at
    the length of real scans: against a numpy fp64 model (Y = X mask H with a real DC / Nyquist bin, amplitudes
    |X mask H|, phases of X), odd trace count (the last pair has one trace)"""
    emu.emu_allow_f(1); emu.emu_allow_p(1)
    emu.emu_set_p_pairs(pairs)
    try:
        assert emu.emu_family(nt) == 6
        nx, ny = 3, 7
        time = synth.make_time(nt)
        cube = synth.make_traces(np.arange(nx * ny) + 5, max(nt, 1024))[:, :nt].reshape(nx, ny, nt).copy()
        chain = synth.default_chain(time)
        npix, nf = nx * ny, nt // 2 + 1
        H = _wiener_cmask(time, nf)
        fft = np.zeros((npix, nf, 2), np.float32); amp = np.zeros((npix, nf), np.float32)
        ph = np.zeros((npix, nf), np.float32); out = np.zeros((npix, nt), np.float32); img = np.zeros(npix, np.float32)
        rc = emu.emu_pipeline_ex(nt, C.c_size_t(npix), _p(cube), _p(chain["w_pre"]), _p(chain["fd_mask"]), _p(H),
                                 _p(chain["w_post"]), _p(fft), _p(amp), _p(ph), _p(out), _p(img))
        assert rc == 0
        X = np.fft.rfft(cube.reshape(npix, nt).astype(np.float64) * chain["w_pre"].astype(np.float64), axis=1)
        Y = X * ((H[:, 0].astype(np.float64) + 1j * H[:, 1]) * chain["fd_mask"])
        a_ref = np.abs(Y)
        Y[:, 0] = Y[:, 0].real
        if nt % 2 == 0:
            Y[:, -1] = Y[:, -1].real
        t_ref = np.fft.irfft(Y, n=nt, axis=1) * chain["w_post"]
        got = fft[..., 0] + 1j * fft[..., 1]
        assert np.abs(got - Y).max() / np.abs(Y).max() < 1e-5
        assert np.all(fft[:, 0, 1] == 0) and (nt % 2 or np.all(fft[:, -1, 1] == 0))
        assert np.abs(amp - a_ref).max() / a_ref.max() < 1e-5
        assert np.abs(out - t_ref).max() / np.abs(t_ref).max() < 1e-5
        assert np.abs(img - (t_ref ** 2).sum(1)).max() / (t_ref ** 2).sum(1).max() < 1e-5
        ref = ob.run_pipeline(cube, time, chain)   # phases are those of X
        d = ph.reshape(ref["phases"].shape) - ref["phases"]
        strong = ref["amplitudes"] > 0.05 * ref["amplitudes"].max(axis=-1, keepdims=True)
        d = d - 2 * np.pi * np.round(d / (2 * np.pi))
        assert np.abs(d[strong]).max() < 3e-3
    finally:
        emu.emu_set_p_pairs(0)


@pytest.mark.parametrize("nt", [1001, 1000])
def test_mixed_radix_stage_kernels_two_pairs_per_wave(emu, nt):
    """k_p<fwd> / k_p<inv> with two pairs of traces per wave == with one (same butterflies, dealt differently),
    for a trace count that leaves the last unit with 1, 2 or 3 of its 4 traces"""
    emu.emu_allow_f(1); emu.emu_allow_p(1)
    time = synth.make_time(nt)
    wa = ob.apply_window(0, np.ones(nt, np.float32), time, 1.0, 7.0)
    mask = ob.fd_bandpass_window(ob.frequency_axis(time), 0.2, 5.0, 0.1)[0]
    win = ob.td_bandpass_window(time, float(time[0]), float(time[-1]), 0.1)[0]
    try:
        for npix in (9, 10, 11):
            x = np.ascontiguousarray(synth.make_traces(np.arange(npix) + 3, max(nt, 1024))[:, :nt], np.float32)
            res = {}
            for q in (1, 2):
                emu.emu_set_p_pairs(q)
                fft = np.zeros((npix, nt // 2 + 1, 2), np.float32); amp = np.zeros((npix, nt // 2 + 1), np.float32)
                ph = np.zeros_like(amp)
                assert emu.emu_fft_fwd(nt, C.c_size_t(npix), _p(x), _p(wa), None, None, _p(fft), _p(amp), _p(ph), _p(mask)) == 0
                out = np.zeros_like(x); img = np.zeros(npix, np.float32)
                assert emu.emu_fft_inv(nt, C.c_size_t(npix), _p(fft), _p(win), _p(out), _p(img)) == 0
                res[q] = (fft, amp, ph, out, img)
            for a, b in zip(res[1], res[2]):
                assert np.array_equal(a, b)
            assert np.abs(res[1][3]).max() > 0
    finally:
        emu.emu_set_p_pairs(0)
