#!/bin/bash
# Runs on the GPU box (through gpurun): kernel trace + one PMC pass of the deconvolution call
# (scripts/gpu_deconv_timing.py, 128 x 128 x 1001, the reference's defaults).
# Usage: scripts/profile_deconv.sh <tag> [nx ny nt]
set -e -o pipefail
tag=${1:-r01g}
nx=${2:-128}; ny=${3:-128}; nt=${4:-1001}
out=gpurun_out/prof/$tag
mkdir -p "$out"
export TMPDIR=/tmp
CMD="python3 scripts/gpu_deconv_timing.py $nx $ny $nt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- $CMD > "$out/trace.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU \
    --output-format csv -d "$out/sq" -- $CMD > "$out/sq.log" 2>&1
python3 scripts/summarize_prof.py "$out" > "$out/summary.txt" 2>&1 || true
cat "$out/summary.txt"
