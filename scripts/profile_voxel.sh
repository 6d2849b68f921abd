#!/bin/bash
# Runs on the GPU box (through gpurun): kernel trace + PMC passes of the voxel-envelope kernels.
# Usage: scripts/profile_voxel.sh <tag> [nx ny nt]
set -e -o pipefail
tag=${1:-r01v}; shift || true
out=gpurun_out/prof/$tag
mkdir -p "$out"
export TMPDIR=/tmp
CMD="python3 scripts/gpu_voxel_timing.py $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- $CMD > "$out/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- $CMD > "$out/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -- $CMD > "$out/write.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS \
    --output-format csv -d "$out/sq" -- $CMD > "$out/sq.log" 2>&1
python3 scripts/summarize_prof.py "$out" > "$out/summary.txt" 2>&1 || true
cat "$out/summary.txt"
