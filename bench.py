#!/usr/bin/env python3
"""bench.py — traces/s of the full default recompute chain on the synthetic
1024x1024x4096 fp32 cube (BASELINE.json metric), one process per GPU.

A step = one UpdateType::Filter(1) of the reference's data thread (data_thread.rs:1023-1334) served by the
library, cube already resident in HBM:  thz_group_session_recompute =
    on every rank's x-slab:  the fused launch (window -> R2C -> |.|/arg/unwrap -> band-pass -> C2R/nt ->
                             taper -> image) + the pixel-sum passes over amplitudes and phases + avg_fft by
                             linearity from the cached mean trace (one nt-point transform)
    C2  ncclAllReduce of the pixel sums, C1 gather of per-pixel results to rank 0 (--gather small|time|all)
Everything between "step starts" and "results on rank 0" happens inside libthzgpu.so (RCCL called by the
library on its own streams); Python parses arguments, launches, and times.  torch.distributed (gloo) is used
for the launch plumbing only: shipping the RCCL unique id, the barrier and the max over ranks of the timing
contract.  The cube is fixed as N grows (strong scaling: BASELINE.json asks for the same cube at 1, 2, 4 and
8 GPUs); --scaling weak keeps a fixed slab per GPU.

One fixed pentagon is set as a region of interest, so every timed step also takes the region's
masked means, as the reference's ifft stage does for every region with every recompute (math_tools.rs:473-543).

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel (the fused launch) from hipEvents
recorded on the engine's own stream around every timed launch; `cpu_baseline` times the CPU oracle (port of
the reference algorithm, OpenMP over x rows like the reference's rayon split) on a bounded sample on rank 0
at N = 1.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes_per_trace(nt):
    """M_full, SURVEY.md §8d: read trace, write spectrum + |.| + phase + trace + pixel"""
    nf = nt // 2 + 1
    return 4 * nt + 16 * nf + 4 * nt + 4


def measured_traffic(nt, traces_per_launch, wiener=False, sums=False):
    """HBM bytes per launch of the fused kernel from the committed PMC passes
    (profiles/r*_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in their
    own runs, gfx950 x2 correction on FETCH_SIZE).  Counters cannot be read from
    inside this process, so this is the latest stored measurement scaled to the
    launch size; None when no profile matches the trace length."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if ("16,16,8" in d.get("kernel", "") and nt == 4096 and bool(d.get("wiener", False)) == wiener
                and bool(d.get("sums", False)) == sums):
            best = d
    if best is None:
        return None
    return best["hbm_bytes_per_trace"] * traces_per_launch


def cpu_baseline(nt, ny, budget_s):
    """Oracle (port) on host cores: fused default chain + pixel means on a
    bounded slab, repeated until ~budget_s of CPU work has been done."""
    import oracle_binding as ob
    import synth

    cores = ob.max_threads()
    # one x row per task like rayon's split over Axis(0): give every thread rows
    nx_s, ny = max(2 * cores, 16), 64
    tm, cube = synth.make_cube(nx_s, ny, nt)
    chain = synth.oracle_chain(tm)
    ob.run_pipeline(cube[:1], tm, chain)  # warm (plans, page faults)
    done, t_used, passes = 0, 0.0, 0
    while t_used < budget_s and passes < 64:
        t0 = time.perf_counter()
        r = ob.run_pipeline(cube, tm, chain)
        ob.pixel_mean(r["fft"], 2); ob.pixel_mean(r["amplitudes"], 1); ob.pixel_mean(r["phases"], 1)
        t_used += time.perf_counter() - t0
        done += nx_s * ny
        passes += 1
    # The reference does not fuse: every stage returns a new container (clone of all arrays,
    # data_thread.rs:1108-1191).  The same oracle stages run one by one with those deep copies show
    # that cost shape; a few passes only — it is an extra, not the baseline.
    staged_s, staged_n = 0.0, 0
    try:
        freq = chain["frequency"]
        for _ in range(3):
            t0 = time.perf_counter()
            c0 = cube.copy()                                                        # scaling: identity clone
            c1 = (c0 * chain["w_tilt"]).astype(np.float32)                          # Tilt Compensation (0 deg)
            c2, _, _ = ob.td_bandpass(c1, tm, float(tm[0]), float(tm[-1]), 2.0)     # Time Band Pass
            st = ob.fft_stage(c2, tm, 0, 1.0, 7.0)                                  # fft
            f2, a2 = ob.fd_bandpass(st["fft"], st["amplitudes"], freq, 0.2, 5.0, 0.1)  # Frequency Band Pass
            ph2 = st["phases"].copy()
            ob.pixel_mean(f2, 2); ob.pixel_mean(a2, 1); ob.pixel_mean(ph2, 1)       # ifft: means ...
            t3, _ = ob.ifft_stage(f2, nt)                                           # ... and C2R
            t4, _, _ = ob.td_bandpass(t3, tm, float(tm[0]), float(tm[-1]), 0.1)     # Time Band Pass (after)
            ob.intensity(t4)
            staged_s += time.perf_counter() - t0
            staged_n += nx_s * ny
    except Exception:  # the extra never breaks the bench line
        staged_n = 0
    out = {"value": done / t_used, "unit": "traces/s", "cores": cores, "kind": "port",
           "sample": f"{nx_s}x{ny}x{nt} slab of the same synthetic cube x {passes} passes "
                     f"({t_used:.1f} s of CPU work), oracle/thz_oracle.c thz_oracle_pipeline + pixel means, "
                     f"OpenMP {cores} threads"}
    if staged_n:
        out["stage_by_stage_with_copies"] = staged_n / staged_s
    return out


# "ROI: one fixed pentagon" — vertices (x, y) as the reference keeps them, on the 1024 x 1024 grid
# (scaled with the grid otherwise); 23 359 pixels inside by the reference's integer rule, 2.2 % of the image
PENTAGON_1024 = ((300, 200), (520, 260), (600, 480), (420, 620), (240, 460))


def pentagon(nx, ny):
    return np.array([[x * ny // 1024, y * nx // 1024] for x, y in PENTAGON_1024], np.uint64)


def spot_check(sess, tm, first_trace, npix, H=None, n=48):
    """The headline run proves itself: n seeded traces of the step's resident outputs against a numpy fp64
    model of the chain built on the ORACLE's multiplier vectors (synth.oracle_chain; the traces are regenerated
    on the host from the counter-based generator).  Relative max-norm errors per output; outside the timed region."""
    import synth
    import thz_image_explorer_amd as pkg
    nt = tm.size
    rng = np.random.default_rng(0x7A3D2026)
    pix = np.sort(rng.choice(npix, size=min(n, npix), replace=False))
    raw = synth.make_traces(np.uint64(first_trace) + pix.astype(np.uint64), nt).astype(np.float64)
    c = synth.oracle_chain(tm)
    pre = c["w_tilt"].astype(np.float64) * c["w_td_before"] * c["w_fft"]
    X = np.fft.rfft(raw * pre, axis=1)
    m = c["fd_mask"].astype(np.float64) * (1.0 if H is None else (H[:, 0].astype(np.float64) + 1j * H[:, 1]))
    Y = X * m
    amp = np.abs(Y)
    Y[:, 0] = Y[:, 0].real
    if nt % 2 == 0:
        Y[:, -1] = Y[:, -1].real
    data = np.fft.irfft(Y, n=nt, axis=1) * c["w_post"]
    img = (data ** 2).sum(1)
    got = {k: np.stack([sess.download(b, int(p), 1)[0] for p in pix])
           for k, b in (("fft", pkg.BUF_FFT), ("amp", pkg.BUF_AMPLITUDES), ("ph", pkg.BUF_PHASES), ("data", pkg.BUF_DATA), ("img", pkg.BUF_IMG))}
    g_fft = got["fft"][..., 0] + 1j * got["fft"][..., 1]
    err = {"fft": float(np.abs(g_fft - Y).max() / np.abs(Y).max()),
           "amplitudes": float(np.abs(got["amp"] - amp).max() / amp.max()),
           "data": float(np.abs(got["data"] - data).max() / np.abs(data).max()),
           "img": float(np.abs(got["img"].ravel() - img).max() / img.max())}
    # unwrapped phases (of X, as with the band pass): equal up to whole turns where the amplitude carries a phase
    ax = np.abs(X)
    strong = ax > 0.05 * ax.max(1, keepdims=True)
    d = got["ph"].astype(np.float64) - np.unwrap(np.angle(X), axis=1)
    d -= 2 * np.pi * np.round(d / (2 * np.pi))
    err["phases_mod_2pi_where_strong"] = float(np.abs(d)[strong].max())
    err["traces"] = int(pix.size)
    return err


def wiener_multiplier(eng, tm):
    """K13 of BASELINE config 5 from the synthetic reference pulse (the noise-free template of the cube's traces,
    SURVEY §8d): R through the engine's own window + transform, H by thz_host_wiener_filter"""
    import thz_image_explorer_amd as pkg
    nt = tm.size
    nf = nt // 2 + 1
    z = ((tm - tm[0] - 11.0) / 0.35).astype(np.float32)
    ref = (-z * np.exp(-z * z)).astype(np.float32)
    w = pkg.host_fft_window(tm, 0, 1.0, 7.0)
    d_ref = eng.to_device(ref); d_w = eng.to_device(w); d_rf = eng.empty((nf, 2))
    eng.fft(1, d_ref, d_w, None, None, d_rf, None, None, None)
    R = d_rf.download((nf, 2), np.float32)
    for b in (d_ref, d_w, d_rf):
        b.free()
    return pkg.host_wiener_filter(R, 1e-2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nx", type=int, default=1024)
    ap.add_argument("--ny", type=int, default=1024)
    ap.add_argument("--nt", type=int, default=4096)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--gather", choices=["small", "time", "all"], default="small",
                    help="what C1 brings to rank 0 (SURVEY 8e): image + means / + final trace cube / every output")
    ap.add_argument("--wiener", action="store_true",
                    help="BASELINE config 5: the reference-pulse Wiener multiplier (complex, K13) in the fused launch")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-means", action="store_true", help="no pixel means: the fused launch alone")
    ap.add_argument("--no-roi", action="store_true", help="without the region of interest (the fixed pentagon of SURVEY 8d)")
    ap.add_argument("--no-spot-check", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("THZ_BENCH_DEVICE"):   # rehearsal knob: every rank on this device (with THZ_RCCL_LIB = tests/mock_rccl)
        local_rank = int(os.environ["THZ_BENCH_DEVICE"])
    if args.gpus != world and args.gpus > 1:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with WORLD_SIZE={args.gpus} (got WORLD_SIZE={world})")

    import thz_image_explorer_amd as pkg
    from thz_image_explorer_amd import binding
    import synth

    # launch plumbing: the id of the library's RCCL communicator travels over a gloo broadcast
    dist = None
    uid = None
    if world > 1:   # torch only as the launch plumbing of a process-per-GPU run (gloo: rendezvous, barrier, max over ranks)
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        box = [pkg.group_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        uid = box[0]

    nx, ny, nt = args.nx, args.ny, args.nt
    nf = nt // 2 + 1
    nx_tot = nx if args.scaling == "strong" else nx * world
    x0, nx_loc = pkg.host_slab(nx_tot, world, rank)   # contiguous x rows, like rayon over Axis(0)
    npix = nx_loc * ny

    # raises if libthzgpu.so, the GPU or (N > 1) librccl is missing: no fallback
    group = pkg.Group(device=local_rank, rank=rank, world=world, uid=uid)
    eng = group.engine(0)
    tm = synth.make_time(nt)
    gs = pkg.GroupSession(group, nx_tot, ny, tm)
    sess = gs.member(0)
    d_time = eng.to_device(tm)
    eng.set_time_axis(tm)
    eng.synth_cube(gs.member_buffer(0, pkg.BUF_RAW), npix, x0 * ny, d_time)   # this rank's rows of the one synthetic cube
    gs.upload(None, subtract_bias=False)   # image of the raw cube, raw pixel sums (+ their all-reduce): once per file
    cfg = pkg.chain_cfg_default(tm)
    cfg.want_means = 0 if args.no_means else 1
    H_wiener = None
    if args.wiener:
        H_wiener = wiener_multiplier(eng, tm)
        sess.set_fd_filters(None, H_wiener)
    roi_poly = None if args.no_roi else pentagon(nx_tot, ny)
    if roi_poly is not None:
        gs.set_rois([roi_poly])
    gather = {"small": pkg.GATHER_SMALL, "time": pkg.GATHER_TIME, "all": pkg.GATHER_ALL}[args.gather]

    def step():
        gs.recompute(cfg, 1, gather)

    def fence():
        group.sync()   # hipStreamSynchronize on every stream the engine uses on this rank's device
        if dist is not None:
            dist.barrier()
            group.sync()

    for _ in range(args.warmup):
        step()
    fence()
    eng.enable_timing(2)  # deferred hipEvents around each launch, no host waits
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    pipe_ns, pipe_calls = eng.timing_collect(binding.STAGE_PIPELINE)
    mean_ns, mean_calls = eng.timing_collect(binding.STAGE_MEAN)
    roi_ns, roi_calls = eng.timing_collect(binding.STAGE_ROI)
    eng.timing_collect(binding.STAGE_FFT)
    eng.enable_timing(0)
    # the run's own correctness witness, before the extra kernel legs below reuse the output buffers
    spot = roi_out = None
    if rank == 0 and not args.no_spot_check:
        spot = spot_check(sess, tm, x0 * ny, npix, H_wiener)
    if rank == 0 and roi_poly is not None:
        r = gs.roi(0, want=["signal_fft", "signal"])
        roi_out = {"pixels": int(r["count"]), "finite": bool(np.isfinite(r["signal_fft"]).all() and np.isfinite(r["signal"]).all()),
                   "mean_amplitude_max": float(r["signal_fft"].max())}
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)   # a CPU tensor over gloo
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # Beside the headline, outside the timed region, same HIP-event timing on the kernel's own stream:
    #  - the north star's 60 % target is quoted on the fused window + FFT + band-pass kernel (spectrum only,
    #    M_fwd = 8 nt + 8 bytes per trace);
    #  - BASELINE config 5's chain: the complex reference-pulse multiplier inside the same ONE launch.
    fwd = wleg = nosum = None
    fused_sums = (not args.no_means) and not os.environ.get("THZ_NO_FUSED_SUMS") and nt in (1024, 2048, 4096)
    if rank == 0:
        d_raw = gs.member_buffer(0, pkg.BUF_RAW)
        d_fft = gs.member_buffer(0, pkg.BUF_FFT); d_amp = gs.member_buffer(0, pkg.BUF_AMPLITUDES)
        d_ph = gs.member_buffer(0, pkg.BUF_PHASES); d_out = gs.member_buffer(0, pkg.BUF_DATA)
        chain = synth.default_chain(tm)
        d_pre = eng.to_device(chain["w_pre"]); d_fd = eng.to_device(chain["fd_mask"]); d_post = eng.to_device(chain["w_post"])
        eng.enable_timing(2)
        for _ in range(10):
            eng.fft(npix, d_raw, d_pre, None, None, d_fft, None, None, d_fd)
        eng.sync()
        fwd_ns, fwd_calls = eng.timing_collect(binding.STAGE_FFT)
        if fwd_calls:
            fwd_s = fwd_ns / fwd_calls * 1e-9
            fwd_gbs = npix * (8 * nt + 8) / fwd_s / 1e9
            fwd = {"kernel": "k_f<fwd> (window + R2C + band-pass, spectrum only)", "bytes_per_trace": 8 * nt + 8,
                   "avg_launch_ms": fwd_s * 1e3, "achieved": fwd_gbs, "frac": fwd_gbs / HBM_PEAK_GBPS,
                   "launches_timed": fwd_calls}
        if fused_sums and not args.wiener:
            # the same chain without the in-launch pixel sums (round 1's and the first round-2 builds' dominant kernel)
            for _ in range(6):
                eng.pipeline_ex(npix, d_raw, d_pre, d_fd, None, d_post, d_fft, d_amp, d_ph, d_out, None, None)
            eng.sync()
            n_ns, n_calls = eng.timing_collect(binding.STAGE_PIPELINE)
            if n_calls:
                n_s = n_ns / n_calls * 1e-9
                n_gbs = npix * algorithmic_bytes_per_trace(nt) / n_s / 1e9
                nosum = {"kernel": "k_f<pipe> (the fused chain without the pixel sums)", "bytes_per_trace": algorithmic_bytes_per_trace(nt),
                         "avg_launch_ms": n_s * 1e3, "achieved": n_gbs, "frac": n_gbs / HBM_PEAK_GBPS, "launches_timed": n_calls,
                         "traffic": measured_traffic(nt, npix, wiener=False, sums=False)}
        if not args.wiener:
            # BASELINE config 5's chain AS A SESSION RUNS IT: complex multiplier AND the in-launch pixel sums
            d_H = eng.to_device(wiener_multiplier(eng, tm))
            d_sums = eng.empty((2 * nf,)) if fused_sums else None
            # the band pass's own index range (what the session passes along): the kernel stages only those bins of H
            _, band_lo, band_hi = pkg.host_fd_bandpass(pkg.host_frequency_axis(tm), 0.2, 5.0, 0.1)
            for _ in range(6):
                eng.pipeline_ex(npix, d_raw, d_pre, d_fd, d_H, d_post, d_fft, d_amp, d_ph, d_out, None, d_sums,
                                band=(int(band_lo), int(band_hi)))
            eng.sync()
            w_ns, w_calls = eng.timing_collect(binding.STAGE_PIPELINE)
            if w_calls:
                w_s = w_ns / w_calls * 1e-9
                w_gbs = npix * algorithmic_bytes_per_trace(nt) / w_s / 1e9
                wleg = {"kernel": "k_f<pipe, complex multiplier" + (", sums" if fused_sums else "") + "> (BASELINE config 5: reference-pulse "
                                  "Wiener filter in the fused launch" + (", with the in-launch pixel sums a session recompute carries" if fused_sums else "") + ")",
                        "bytes_per_trace": algorithmic_bytes_per_trace(nt), "avg_launch_ms": w_s * 1e3, "achieved": w_gbs,
                        "frac": w_gbs / HBM_PEAK_GBPS, "launches_timed": w_calls,
                        "traffic": measured_traffic(nt, npix, wiener=True, sums=fused_sums)}
        eng.enable_timing(0)

    # sanity: the run produced finite, non-trivial output (checked outside the timed region)
    if rank == 0:
        img_h = gs.download(pkg.BUF_IMG)
        assert np.isfinite(img_h).all() and img_h.max() > 0, "pipeline produced no output"

    if rank == 0:
        total_traces = nx_tot * ny
        value = total_traces * args.steps / dt
        m_full = algorithmic_bytes_per_trace(nt)
        k_avg_s = pipe_ns / max(pipe_calls, 1) * 1e-9
        achieved = npix * m_full / k_avg_s / 1e9 if k_avg_s > 0 else 0.0
        out = {
            "metric": "traces/sec full pipeline on 1024x1024x4096 cube; achieved HBM GB/s %",
            "value": value, "unit": "traces/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{nx_tot}x{ny}x{nt} fp32 synthetic cube (Philox seed 0x7A3D2026), default chain: "
                                   "tilt taper + Time Band Pass + fft window -> R2C -> amplitude/phase/unwrap -> "
                                   "Frequency Band Pass 0.2-5 THz"
                                   + (" x reference-pulse Wiener multiplier (complex)" if args.wiener else "")
                                   + " -> C2R/nt -> Time Band Pass -> intensity image"
                                   + ("" if args.no_means else " + pixel-mean spectra"),
                       "parallelism": f"x-slab tiles, {world} rank(s), {nx_loc}x{ny} traces per GPU"
                                      + (f"; in-library RCCL: all-reduce of pixel sums + gather '{args.gather}' to rank 0"
                                         if world > 1 else ""),
                       "gather": args.gather, "step": "thz_group_session_recompute (UpdateType::Filter(1))",
                       "rccl_ranks": group.world,   # thz_group_world: the communicator size the library saw
                       "roi": (None if roi_poly is None else
                               {"polygon": roi_poly.tolist(), **(roi_out or {}),
                                "ms_per_step": (roi_ns / max(args.steps, 1)) * 1e-6 if roi_calls else None}),
                       "spot_check_rel_err": spot,
                       "kernel_variant": eng.kernel_variant(),
                       "achieved_hbm_pct_whole_step": 100.0 * value / world * m_full / 1e9 / HBM_PEAK_GBPS},
            "roofline": {"bound": "hbm",
                         "kernel": ("k_f<pipe, sums> (fused default chain + amplitude / phase pixel sums in the same launch)"
                                    if fused_sums else "k_f<pipe> (fused default chain)"),
                         "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": measured_traffic(nt, npix, wiener=args.wiener, sums=fused_sums),
                         "bytes_per_trace": m_full, "traces_per_launch": npix,
                         "avg_launch_ms": k_avg_s * 1e3, "launches_timed": pipe_calls,
                         "pixel_sum_passes_ms_per_step": (mean_ns / max(args.steps, 1)) * 1e-6 if mean_calls else None,
                         "fused_forward_kernel": fwd, "wiener_leg": wleg, "fused_chain_without_sums": nosum,
                         "note": ("the dominant kernel carries the pixel sums since late round 2 (they were a 3.1 ms second pass over "
                                  "16 GiB): compare `fused_chain_without_sums` with earlier rounds' roofline.frac, and `config."
                                  "achieved_hbm_pct_whole_step` for the step as a whole") if fused_sums else None},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(nt, ny, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()   # rank 0 ran its extra kernel legs meanwhile: the communicators go down together
    gs.close()
    group.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
