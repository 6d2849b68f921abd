"""Generates the golden fixtures under tests/golden/ (run in the authoring
container; the GPU box only reads the committed .npz files).

Nothing here imports or runs reference code (the reference is Rust and cannot
be built here).  Expected outputs come from numpy in float64
(`numpy.fft`, third-party pocketfft) and from formulas re-stated below in
plain numpy/Python ints, independently of oracle/thz_oracle.c, so that the
oracle itself can be checked against them.

Fixtures
  fft_vectors.npz      seeded synthetic traces at Nt in {128,1000,1001,1024,4096}
                       x 5 window types (4 traces each) -> windowed trace, rfft, |.|, unwrapped
                       phase, band-passed spectrum, irfft (all float64 truth)
  unit_signals.npz     the input signals of the reference's own unit tests
                       (math_tools.rs:843-897, band_pass_fd.rs:475-567,
                       band_pass_td_before_fft.rs:390-443) with fp64 expectations
  knife_edge.npz       16 real traces (Nt = 1001) read with h5dump from the
                       reference's sample_data/example_beam_width (data only)
  roi_masks.npz        integer ROI masks for 8 polygons computed with exact
                       Python ints under the release-mode wrapping rule
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"

PI32 = np.float32(np.pi)


# ---- formulas re-stated in numpy (float32 where the reference is f32) -----
def blackman_window(n, m):
    with np.errstate(all="ignore"):
        n = np.float32(n); m = np.float32(m)
        res = (np.float32(0.42) - np.float32(0.5) * np.cos(np.float32(2.0) * PI32 * n / m, dtype=np.float32)
               + np.float32(0.08) * np.cos(np.float32(4.0) * PI32 * n / m, dtype=np.float32))
    if np.isnan(res):
        return np.float32(1.0)
    return np.float32(min(max(res, np.float32(0.0)), np.float32(1.0)))


def adapted_blackman(time, lo, hi):
    time = np.asarray(time, np.float32)
    lo = np.float32(lo); hi = np.float32(hi)
    w = np.ones(time.size, np.float32)
    t0, tn = time[0], time[-1]
    for i, t in enumerate(time):
        if t <= lo + t0:
            w[i] = blackman_window(t - t0, np.float32(2.0) * lo)
        elif t >= tn - hi:
            w[i] = blackman_window(t - (tn - hi * np.float32(2.0)), np.float32(2.0) * hi)
    return w


def full_window(kind, time):
    time = np.asarray(time, np.float32)
    tau = ((time - time.min()) / (time.max() - time.min())).astype(np.float32)
    c = lambda k: np.cos(np.float32(k) * PI32 * tau, dtype=np.float32)
    if kind == 1:
        return (np.float32(0.42) - np.float32(0.5) * c(2.0) + np.float32(0.08) * c(4.0)).astype(np.float32)
    if kind == 2:
        return (np.float32(0.5) * (np.float32(1.0) - c(2.0))).astype(np.float32)
    if kind == 3:
        return (np.float32(0.54) - np.float32(0.46) * c(2.0)).astype(np.float32)
    if kind == 4:
        return (np.float32(1.0) - np.float32(1.93) * c(2.0) + np.float32(1.29) * c(4.0)
                - np.float32(0.388) * c(6.0) + np.float32(0.028) * c(8.0)).astype(np.float32)
    raise ValueError(kind)


def unwrap64(ph):
    """numpy_unwrap recurrence (math_tools.rs:226-238) in float64"""
    out = np.empty_like(ph)
    out[0] = ph[0]
    for i in range(1, ph.size):
        d = ph[i] - ph[i - 1]
        if d > np.pi:
            d -= 2 * np.pi
        elif d < -np.pi:
            d += 2 * np.pi
        out[i] = out[i - 1] + d
    return out


def fd_mask(freq, low, high, width):
    freq = np.asarray(freq, np.float32)
    safe_low = np.float32(max(low, 0.0))
    safe_high = np.float32(min(high, float(freq[-1])))
    idx = np.nonzero(freq >= safe_low)[0]
    lower = int(idx[0]) if idx.size else 0
    idx = np.nonzero(freq <= safe_high)[0]
    upper = int(idx[-1]) + 1 if idx.size else freq.size
    m = np.zeros(freq.size, np.float32)
    if upper > lower:
        m[lower:upper] = adapted_blackman(freq[lower:upper], width, width)
    return m, lower, upper


def td_mask(time, low, high, width):
    time = np.asarray(time, np.float32)
    low = max(low, float(time[0])); high = min(high, float(time[-1]))
    idx = np.nonzero(time >= np.float32(low))[0]
    lower = int(idx[0]) if idx.size else 0
    idx = np.nonzero(time >= np.float32(high))[0]
    upper = int(idx[0]) if idx.size else max(time.size - 1, 0)
    upper = min(max(upper, lower + 1), time.size)
    m = np.zeros(time.size, np.float32)
    m[lower:upper] = adapted_blackman(time[lower:upper], width, width)
    return m, lower, upper


# ---- fixtures ------------------------------------------------------------
def gen_fft_vectors():
    import synth
    out = {}
    for nt in (128, 1000, 1001, 1024, 4096):
        time = synth.make_time(nt)
        freq = (np.arange(nt // 2 + 1, dtype=np.float32) / (time[-1] - time[0])).astype(np.float32)
        traces = synth.make_traces(np.arange(4) + 1000 * nt, nt)
        out[f"nt{nt}_time"] = time
        out[f"nt{nt}_freq"] = freq
        out[f"nt{nt}_raw"] = traces
        m, lo, up = fd_mask(freq, 0.2, 5.0, 0.1)
        out[f"nt{nt}_fdmask"] = m
        out[f"nt{nt}_fd_idx"] = np.array([lo, up])
        for kind in range(5):
            w = adapted_blackman(time, 1.0, 7.0) if kind == 0 else full_window(kind, time)
            xw = (traces * w[None, :]).astype(np.float32)      # f32 multiply like the reference
            X = np.fft.rfft(xw.astype(np.float64), axis=1)
            out[f"nt{nt}_w{kind}_window"] = w
            out[f"nt{nt}_w{kind}_windowed"] = xw
            out[f"nt{nt}_w{kind}_fft"] = X
            if kind == 0:
                out[f"nt{nt}_w0_amp"] = np.abs(X)
                out[f"nt{nt}_w0_phase"] = np.stack([unwrap64(np.angle(x)) for x in X])
                Xb = X * m[None, :].astype(np.float64)
                out[f"nt{nt}_w0_irfft_bp"] = np.fft.irfft(Xb, n=nt, axis=1)
    np.savez_compressed(os.path.join(OUT, "fft_vectors.npz"), **out)


def gen_unit_signals():
    out = {}
    # math_tools.rs:843-897 test_fft_roundtrip
    n = 128
    tt = np.arange(n, dtype=np.float32) / np.float32(n)
    sig = (np.sin(np.float32(2.0) * PI32 * np.float32(3) * tt, dtype=np.float32)
           + np.float32(0.5) * np.cos(np.float32(2.0) * PI32 * np.float32(7) * tt, dtype=np.float32))
    out["roundtrip_signal"] = sig.astype(np.float32)
    out["roundtrip_time"] = np.linspace(0.0, 1.0, n, dtype=np.float32)
    out["roundtrip_fft"] = np.fft.rfft(sig.astype(np.float64))
    # band_pass_fd.rs:475-567: 1x1x256 sine at bin 9, freq i/50, pass +-2 bins, width 0
    n = 256
    k = 9
    tt = np.arange(n, dtype=np.float32)
    sig = np.sin(np.float32(2.0) * PI32 * np.float32(k) * tt / np.float32(n), dtype=np.float32)
    out["fd_signal"] = sig
    out["fd_freq"] = (np.arange(n // 2 + 1, dtype=np.float32) / np.float32(50.0)).astype(np.float32)
    # band_pass_td_before_fft.rs:390-443: 1x1x256, low .25, high .55, width 0
    out["td_time"] = np.linspace(0.0, 1.0, n, dtype=np.float32)
    out["td_signal"] = np.sin(np.float32(2.0) * PI32 * np.float32(5) * out["td_time"], dtype=np.float32)
    m, lo, up = td_mask(out["td_time"], 0.25, 0.55, 0.0)
    out["td_mask"] = m
    out["td_idx"] = np.array([lo, up])
    np.savez_compressed(os.path.join(OUT, "unit_signals.npz"), **out)


def gen_knife_edge_thz():
    """tests/golden/knife_edge_2groups.thz: two measurement groups of the reference's real
    sample file, copied verbatim with h5copy (a data file for the dotTHz reader tests)"""
    f = os.path.join(REF, "sample_data/example_beam_width/measurement_x/data/1750085285.8557956_data.thz")
    out = os.path.join(OUT, "knife_edge_2groups.thz")
    if os.path.exists(out):
        os.remove(out)
    for g in ("Beam Width Measurement x=-0.10", "Beam Width Measurement x=-0.20"):
        subprocess.run(["/opt/conda/bin/h5copy", "-i", f, "-o", out, "-s", "/" + g, "-d", "/" + g], check=True)


def gen_knife_edge():
    f = os.path.join(REF, "sample_data/example_beam_width/measurement_x/data/1750085285.8557956_data.thz")
    ls = subprocess.run(["/opt/conda/bin/h5ls", "-r", f], stdout=subprocess.PIPE, text=True, check=True).stdout
    paths = []
    for line in ls.splitlines():
        if " Dataset " in line:
            paths.append(line.split(" Dataset ")[0].rstrip().replace("\\ ", " "))
    paths = paths[:: max(len(paths) // 16, 1)][:16]
    traces, time = [], None
    for p in paths:
        with tempfile.NamedTemporaryFile(suffix=".bin") as tf:
            subprocess.run(["/opt/conda/bin/h5dump", "-d", p, "-b", "LE", "-o", tf.name, f],
                           stdout=subprocess.DEVNULL, check=True)
            a = np.fromfile(tf.name, dtype="<f4").reshape(-1, 2)
        time = a[:, 0].copy()
        traces.append(a[:, 1].copy())
    np.savez_compressed(os.path.join(OUT, "knife_edge.npz"), time=time.astype(np.float32),
                        traces=np.stack(traces).astype(np.float32), groups=np.array(paths))


def pip_wrapping(x, y, poly):
    """point_in_polygon (math_tools.rs:574-591) with exact ints mod 2^64"""
    M = 1 << 64
    inside = False
    panic = False
    j = len(poly) - 1
    for i in range(len(poly)):
        xi, yi = poly[i]
        xj, yj = poly[j]
        if (yi > y) != (yj > y):
            if xj < xi or y < yi or yj < yi:
                panic = True
            num = (((xj - xi) % M) * ((y - yi) % M)) % M
            den = (yj - yi) % M
            rhs = (num // den + xi) % M
            if x < rhs:
                inside = not inside
        j = i
    return inside, panic


def roi_mask_exact(poly, scaling, shape0, shape1):
    poly = [(x // scaling, y // scaling) for x, y in poly]
    x_size, y_size = shape1, shape0
    mask = np.zeros((shape0, shape1), np.uint8)
    xs = [p[0] for p in poly]; ys = [p[1] for p in poly]
    x_min = min(min(xs), x_size - 1); x_max = min(max(xs), x_size - 1)
    y_min = min(min(ys), y_size - 1); y_max = min(max(ys), y_size - 1)
    panic = False
    for y in range(y_min, y_max + 1):
        for x in range(x_min, x_max + 1):
            ins, p = pip_wrapping(x, y, poly)
            panic |= p
            mask[y, x] = 1 if ins else 0
    return mask, panic


def gen_roi_masks():
    polys = {
        "convex_ccw": [(3, 2), (20, 4), (27, 18), (12, 29), (2, 15)],
        "convex_cw": [(2, 15), (12, 29), (27, 18), (20, 4), (3, 2)],
        "concave": [(2, 2), (28, 2), (28, 28), (15, 10), (2, 28)],
        "touch_border": [(0, 0), (31, 0), (31, 31), (0, 31)],
        "triangle": [(5, 5), (25, 8), (10, 27)],
        "outside_clamped": [(20, 20), (200, 25), (180, 300), (25, 150)],
        "thin": [(4, 4), (30, 5), (30, 6), (4, 6)],
        "pentagon": [(40, 10), (200, 30), (240, 100), (120, 125), (20, 80)],
    }
    out = {}
    for name, poly in polys.items():
        for (s0, s1) in ((32, 32), (129, 257)):
            for scaling in (1, 2):
                m, panic = roi_mask_exact(poly, scaling, s0, s1)
                key = f"{name}_{s0}x{s1}_s{scaling}"
                out[key + "_mask"] = m
                out[key + "_panic"] = np.array(panic)
        out[name + "_poly"] = np.array(poly, np.uint64)
    np.savez_compressed(os.path.join(OUT, "roi_masks.npz"), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    gen_fft_vectors()
    gen_unit_signals()
    gen_roi_masks()
    if os.path.isdir(REF):
        gen_knife_edge()
        gen_knife_edge_thz()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
