#!/bin/bash
# Runs on the GPU box (through gpurun): kernel trace + SQ counter pass of the fused chain at one shape.
# Usage: scripts/profile_pipe.sh <tag> <nx> <ny> <nt>     -> gpurun_out/prof/<tag>/{trace,sq,sq2}/
set -e -o pipefail
tag=$1; nx=$2; ny=$3; nt=$4
out=gpurun_out/prof/$tag
mkdir -p "$out"
export TMPDIR=/tmp THZ_ONLY=pipeline
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 scripts/gpu_pipe_timing.py $nx $ny $nt > "$out/trace.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS \
    --output-format csv -d "$out/sq" -- python3 scripts/gpu_pipe_timing.py $nx $ny $nt > "$out/sq.log" 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS \
    --output-format csv -d "$out/sq2" -- python3 scripts/gpu_pipe_timing.py $nx $ny $nt > "$out/sq2.log" 2>&1 || true
python3 - "$out" > "$out/summary.txt" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("sq", "sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == list(acc[k].keys())[0]: n[k] += 1
    for k, c in acc.items():
        if "k_p<" in k or "k_f<" in k or "k_fb" in k:
            print(sub, k, "dispatches", n[k])
            for name, v in c.items():
                print(f"    {name:28s} {v / max(n[k], 1):16.0f} per dispatch")
for f in glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_p<" in r["Name"] or "k_f<" in r["Name"] or "k_fb" in r["Name"]:
            print("trace", r["Name"][:70], "calls", r["Calls"], "avg ns", r["AverageNs"])
PY
cat "$out/summary.txt"
