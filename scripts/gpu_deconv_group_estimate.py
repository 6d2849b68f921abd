"""Developer tool: what the Deconvolution stage would take on N GPUs, from phase times measured on ONE.
The group runs the stage in phases (csrc/group_api.cpp): A transform + band energies of the rank's rows (per pixel:
1/N of the single-GPU time), B the Richardson-Lucy iterations of the rank's bands over the whole image (bands dealt
out by cost, contiguous ranges), C recombination of the rank's rows (1/N).  B is measured here per rank by running
thz_deconvolve with that rank's band range on the whole image and reading the "iterations" line of THZ_DEBUG_TIMING;
the exchanges are 2 x n_filters x Nx x Ny floats per rank.  No fabric involved: an estimate, not a measurement."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def worker(nx, ny, nt, ranges):
    import numpy as np
    import thz_image_explorer_amd as pkg
    from test_gpu_deconv import _bar_target_cube
    z = np.load(os.path.join(ROOT, "tests", "golden", "psf_sample.npz"))
    psf = pkg.psf_from_npz(z)
    tm, cube = _bar_target_cube(nx, ny, nt)
    eng = pkg.Engine(0); eng.set_time_axis(tm)
    d_in = eng.to_device(cube); d_out = eng.empty((nx * ny, nt)); d_img = eng.empty((nx * ny,))
    for b0, b1 in ranges:
        cfg = pkg.DeconvCfg(500, 25, 0.1, 10.0, 0.5, b0, b1)
        for k in range(2):
            sys.stderr.write(f"RANGE {b0} {b1} call {k}\n"); sys.stderr.flush()
            eng.deconvolve(psf, cfg, nx, ny, 0.5, 0.5, d_in, d_out, d_img); eng.sync()

def run(nx, ny, nt, ranges, bands=False):
    """-> {(b0, b1): phase times of the second call}, band table [(n_iter, tiles)] of the first range"""
    env = dict(os.environ, THZ_DEBUG_TIMING="1")
    if bands:
        env["THZ_DEBUG_BANDS"] = "1"
    arg = ",".join(f"{a}:{b}" for a, b in ranges)
    r = subprocess.run([sys.executable, __file__, "worker", str(nx), str(ny), str(nt), arg], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    if r.returncode:
        raise RuntimeError(r.stderr[-2000:])
    out, table = {}, []
    for seg in r.stderr.split("RANGE ")[1:]:
        head, _, body = seg.partition("\n")
        b0, b1, _, k = head.split()
        ph = {m.group(1).strip(): float(m.group(2)) for m in re.finditer(r"thz_deconvolve: ([a-z,+ ]+?)\s+([0-9.]+) ms", body)}
        if k == "1":
            out[(int(b0), int(b1))] = ph
        if bands and not table:
            table = [(int(m.group(1)), int(m.group(2))) for m in re.finditer(r"n_iter\s+(\d+)\s+tiles\s+(\d+)", body)]
    return out, table

if __name__ == "__main__":
    if sys.argv[1:2] == ["worker"]:
        worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), [tuple(int(v) for v in r.split(":")) for r in sys.argv[5].split(",")])
        sys.exit(0)
    nx, ny, nt = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (512, 512, 1001)))
    res, table = run(nx, ny, nt, [(0, 0)], bands=True)
    ph = res[(0, 0)]
    nb = len(table)
    a, b, c = ph["transform, band energies"], ph["iterations"], ph["gains, recombination"]
    print(f"{nx}x{ny}x{nt}, 500 iterations, {nb} bands, one GPU: A transform + energies {a:.2f} ms, B iterations {b:.2f} ms, C gains + recombination {c:.2f} ms")
    xch = 2 * nb * nx * ny * 4 / 153e9 * 1e3  # one xGMI link's worth, an upper bound per rank
    alpha, beta = 13.2, 0.0204
    cost = lambda lo, hi: alpha * max([table[k][0] for k in range(lo, hi)] + [0]) + beta * sum(table[k][0] * table[k][1] for k in range(lo, hi))
    plans = {}
    for world in (2, 4, 8):
        # the ranges the library deals out (csrc/group_api.cpp: dynamic programme over the cut points on the model
        # alpha x longest band's iterations + beta x sum of iterations x tiles)
        best = [[1e300] * (nb + 1) for _ in range(world + 1)]
        cut = [[0] * (nb + 1) for _ in range(world + 1)]
        best[0][0] = 0.0
        for q in range(1, world + 1):
            for hi in range(nb + 1):
                for lo in range(hi + 1):
                    if best[q - 1][lo] < 1e300:
                        v = max(best[q - 1][lo], cost(lo, hi))
                        if v < best[q][hi]:
                            best[q][hi], cut[q][hi] = v, lo
        bnd, hi = [nb], nb
        for q in range(world, 0, -1):
            hi = cut[q][hi]
            bnd.insert(0, hi)
        plans[world] = bnd
    todo = sorted({(bnd[q], bnd[q + 1]) for bnd in plans.values() for q in range(len(bnd) - 1) if bnd[q + 1] > bnd[q]})
    res, _ = run(nx, ny, nt, todo)
    for world, bnd in plans.items():
        times = [res[(bnd[q], bnd[q + 1])]["iterations"] if bnd[q + 1] > bnd[q] else 0.0 for q in range(world)]
        model = [round(cost(bnd[q], bnd[q + 1]) * 1e-3, 2) for q in range(world)]
        est = a / world + max(times) + c / world + xch
        print(f"N = {world}: band ranges {[(bnd[q], bnd[q + 1]) for q in range(world)]}, iterations per rank "
              f"{[round(t, 2) for t in times]} ms (model {model}) -> A/N {a / world:.2f} + max B {max(times):.2f} + C/N {c / world:.2f} + exchanges <= {xch:.2f} = {est:.1f} ms "
              f"(one GPU: {a + b + c:.1f})", flush=True)
