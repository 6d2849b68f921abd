#!/usr/bin/env python3
"""bench.py — traces/s of the full default recompute chain on the synthetic
1024x1024x4096 fp32 cube (BASELINE.json metric), one process per GPU.

A step = one pass of the hot path over the rank's x-slab of the cube, cube
already resident in HBM:
    thz_pipeline   (window -> R2C -> |.|/arg/unwrap -> band-pass -> C2R/nt -> taper -> image)
    thz_pixel_sum  x3 (avg_fft, avg_signal_fft, avg_phase_fft partials, math_tools.rs:421-440)
    N > 1: RCCL all-reduce of the partial sums + gather of the image slabs to rank 0
The cube is fixed as N grows (strong scaling: BASELINE.json asks for the same
cube at 1, 2, 4 and 8 GPUs); use --scaling weak for a fixed slab per GPU.

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel
(k_pipeline) from hipEvents recorded on the engine's own stream around every
timed launch; `cpu_baseline` times the CPU oracle (port of the reference
algorithm, OpenMP over x rows like the reference's rayon split) on a bounded
sample on rank 0 at N=1.
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


def measured_traffic(nt, traces_per_launch):
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
        if "16,16,8" in d.get("kernel", "") and nt == 4096:
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
    chain = synth.default_chain(tm)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nx", type=int, default=1024)
    ap.add_argument("--ny", type=int, default=1024)
    ap.add_argument("--nt", type=int, default=4096)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-means", action="store_true", help="time thz_pipeline alone")
    ap.add_argument("--dist-backend", default="nccl",
                    help="nccl (= RCCL, the default) or gloo to rehearse the N>1 path on a one-GPU box "
                         "(ranks then share device LOCAL_RANK %% device_count)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with WORLD_SIZE={args.gpus} "
                             f"(got WORLD_SIZE={world})")
    import torch

    from thz_image_explorer_amd import Engine, binding, shard
    import synth

    dist = None
    force_dist = os.environ.get("THZ_BENCH_FORCE_DIST") == "1"  # rehearse RCCL with a single rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local_rank)
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    nx, ny, nt = args.nx, args.ny, args.nt
    nf = nt // 2 + 1
    if args.scaling == "strong":
        x0, nx_loc = shard.slab(nx, world, rank)  # contiguous x rows, like rayon over Axis(0)
        nx_tot = nx
    else:
        nx_loc, x0 = nx, rank * nx
        nx_tot = nx * world
    npix = nx_loc * ny

    eng = Engine(local_rank)  # raises if libthzgpu.so or the GPU is missing: no fallback
    tm = synth.make_time(nt)
    eng.set_time_axis(tm)
    chain = synth.default_chain(tm)
    d_time = eng.to_device(tm)
    d_raw = eng.empty((npix, nt))
    eng.synth_cube(d_raw, npix, x0 * ny, d_time)
    d_pre = eng.to_device(chain["w_pre"]); d_fd = eng.to_device(chain["fd_mask"]); d_post = eng.to_device(chain["w_post"])
    d_fft = eng.empty((npix, nf, 2)); d_amp = eng.empty((npix, nf)); d_ph = eng.empty((npix, nf))
    d_out = eng.empty((npix, nt))
    # small products live in torch tensors so RCCL can move them
    t_img = torch.empty((nx_loc, ny), dtype=torch.float32, device=dev)
    t_sums = torch.empty(4 * nf, dtype=torch.float32, device=dev)  # [fft re/im interleaved | amp | phase]
    ext = torch.cuda.ExternalStream(eng.stream, device=dev)
    p_img, p_sums = t_img.data_ptr(), t_sums.data_ptr()

    def step():
        eng.pipeline(npix, d_raw, d_pre, d_fd, d_post, d_fft, d_amp, d_ph, d_out, p_img)
        if not args.no_means:
            eng.pixel_sum(npix, nf, 2, d_fft, p_sums)
            eng.pixel_sum(npix, nf, 1, d_amp, p_sums + 8 * nf)
            eng.pixel_sum(npix, nf, 1, d_ph, p_sums + 12 * nf)
        if dist is not None:
            with torch.cuda.stream(ext):  # collectives ordered after the kernels, no host sync
                if not args.no_means:
                    shard.all_reduce_sums(t_sums, dist)      # C2
                shard.gather_image(t_img, nx_tot, dist)      # C1

    def fence():
        eng.sync()
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier(device_ids=[local_rank]) if args.dist_backend == "nccl" else dist.barrier()
            torch.cuda.synchronize(dev)

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
    eng.enable_timing(0)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # The north star's 60 % target is quoted on the fused window + FFT + band-pass kernel (spectrum only,
    # M_fwd = 8 nt + 8 bytes per trace): measured beside the headline, outside the timed region,
    # with the same HIP-event timing on the kernel's own stream.
    fwd = None
    if rank == 0:
        eng.enable_timing(2)
        for _ in range(10):
            eng.fft(npix, d_raw, d_pre, None, None, d_fft, None, None, d_fd)
        eng.sync()
        fwd_ns, fwd_calls = eng.timing_collect(binding.STAGE_FFT)
        eng.enable_timing(0)
        if fwd_calls:
            fwd_s = fwd_ns / fwd_calls * 1e-9
            fwd_gbs = npix * (8 * nt + 8) / fwd_s / 1e9
            fwd = {"kernel": "k_f<fwd> (window + R2C + band-pass, spectrum only)", "bytes_per_trace": 8 * nt + 8,
                   "avg_launch_ms": fwd_s * 1e3, "achieved": fwd_gbs, "frac": fwd_gbs / HBM_PEAK_GBPS,
                   "launches_timed": fwd_calls}

    # sanity: the run produced finite, non-trivial output (checked outside the timed region)
    img_h = t_img.cpu().numpy()
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
                                   "Frequency Band Pass 0.2-5 THz -> C2R/nt -> Time Band Pass -> intensity image"
                                   + ("" if args.no_means else " + pixel-mean spectra"),
                       "parallelism": f"x-slab tiles, {world} rank(s), {nx_loc}x{ny} traces per GPU"
                                      + ("; RCCL all-reduce of mean partials + image gather" if world > 1 else ""),
                       "kernel_variant": eng.kernel_variant(),
                       "achieved_hbm_pct_whole_step": 100.0 * value / world * m_full / 1e9 / HBM_PEAK_GBPS},
            "roofline": {"bound": "hbm", "kernel": "k_pipeline", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": measured_traffic(nt, npix),
                         "bytes_per_trace": m_full, "traces_per_launch": npix,
                         "avg_launch_ms": k_avg_s * 1e3, "launches_timed": pipe_calls,
                         "fused_forward_kernel": fwd},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(nt, ny, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
