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
python3 scripts/pmc_summary.py "$out" > "$out/summary.txt"
cat "$out/summary.txt"
