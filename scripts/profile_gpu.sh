#!/bin/bash
# Runs on the GPU box (through gpurun): kernel trace + PMC passes of bench.py.
# Usage: scripts/profile_gpu.sh <tag> [bench args...]
# Output: gpurun_out/prof/<tag>/{trace,fetch,write,sq}/ and a text summary.
set -e -o pipefail
tag=${1:-r01}; shift || true
out=gpurun_out/prof/$tag
mkdir -p "$out"
export TMPDIR=/tmp
BENCH="python3 bench.py --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- $BENCH --steps 10 --warmup 2 > "$out/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- $BENCH --steps 2 --warmup 1 > "$out/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -- $BENCH --steps 2 --warmup 1 > "$out/write.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS \
    --output-format csv -d "$out/sq" -- $BENCH --steps 2 --warmup 1 > "$out/sq.log" 2>&1
python3 scripts/summarize_prof.py "$out" > "$out/summary.txt" 2>&1 || true
cat "$out/summary.txt"
