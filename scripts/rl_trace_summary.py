"""Developer tool: per-chain summary of the Richardson-Lucy launches in a rocprofv3 kernel trace CSV of
scripts/gpu_deconv_timing.py (which call: argv[2], default 1 = the timed 500-iteration call)."""
import collections, csv, sys
import numpy as np
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
call = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kind = lambda n: "sep" if "k_rl_step_sep" in n else "wide" if "ILb1" in n else "narrow"
rl = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Queue_Id"], kind(r["Kernel_Name"]))
            for r in rows if "k_rl_step_tiled" in r["Kernel_Name"] or "k_rl_step_sep" in r["Kernel_Name"])
inits = sorted(int(r["Start_Timestamp"]) for r in rows if "k_rl_init" in r["Kernel_Name"])
gains = sorted(int(r["Start_Timestamp"]) for r in rows if "k_dc_gain" in r["Kernel_Name"])
print("init -> gain of every call (us):", [round((g - i) / 1e3) for i, g in zip(inits, gains)])
i0, g0 = inits[call], gains[call]
sel = [x for x in rl if i0 < x[0] < g0]
byq = collections.defaultdict(list)
for x in sel:
    byq[(x[3], x[2], x[4])].append(x)
for k, v in sorted(byq.items()):
    d = np.array([(b - a) / 1e3 for a, b, *_ in v])
    st = np.array([a for a, *_ in v]); en = np.array([b for _, b, *_ in v])
    gap = (st[1:] - en[:-1]) / 1e3
    print(f"queue {k[0]} grid {k[1]:4d} {k[2]:6s}: {len(v):4d} launches, avg {d.mean():6.2f} us, busy {d.sum() / 1e3:6.2f} ms, "
          f"first at {(v[0][0] - i0) / 1e3:7.1f} us, last end {(v[-1][1] - i0) / 1e3:8.1f} us; gaps: median {np.median(gap):5.2f}, sum {gap.sum() / 1e3:6.2f} ms, >100 us: {(gap > 100).sum()}")
    # by batch of 64 launches
    line = []
    for lo in range(0, len(v), 64):
        line.append(f"{d[lo:lo + 64].mean():.1f}")
    print("      avg us per batch:", " ".join(line))
