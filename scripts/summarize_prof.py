"""Summarises rocprofv3 CSV output of scripts/profile_gpu.sh into text."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(sub, pat):
    r = glob.glob(os.path.join(root, sub, "**", pat), recursive=True)
    return r[0] if r else None


def short(n):
    n = n.split("(")[0]
    return n.replace("thz::", "").replace("void ", "")[:70]


f = find("trace", "*kernel_stats.csv")
if f:
    print("== rocprofv3 --kernel-trace --stats ==")
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print(f"{short(r['Name']):70s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:10.1f} us  "
              f"total {float(r['TotalDurationNs'])/1e6:9.2f} ms  {float(r['Percentage']):5.1f} %")
f = find("trace", "*kernel_trace.csv")
if f:
    d = defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        d[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        meta[k] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"),
                   r.get("Scratch_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"))
    print("\n== per-kernel dispatch durations (kernel_trace) ==")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:10]:
        v2 = sorted(v)
        print(f"{k:70s} n={len(v):4d} median {v2[len(v2)//2]/1e3:10.1f} us  min {v2[0]/1e3:10.1f}  "
              f"vgpr/agpr/sgpr/lds/scratch/grid/wg = {meta[k]}")
for sub, names in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]),
                   ("sq", ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                           "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VALU", "SQ_INSTS_LDS"])):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"\n== PMC pass '{sub}' (per dispatch, mean) ==")
    for k, c in acc.items():
        if not any(n in c for n in names):
            continue
        parts = []
        for n in names:
            if n in c:
                parts.append(f"{n}={sum(c[n])/len(c[n]):.4g}")
        print(f"{k:60s} " + " ".join(parts))
