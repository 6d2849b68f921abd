#!/bin/bash
# Runs on the GPU box (through gpurun): bench.py as the driver launches it for N > 1 — torch.distributed.run, one
# process per rank — with every rank on device 0 and tests/mock_rccl in RCCL's place (THZ_BENCH_DEVICE=0,
# THZ_RCCL_LIB).  It rehearses the launch plumbing, the slab partition, the library's sequence of collectives and the
# JSON line; it says nothing about a fabric and its numbers are not scaling numbers.  The box allows six processes on
# its card (the launcher counts), so four ranks is what a rehearsal holds (eight ranks: the driver's SCALE run on an 8-GPU node).
set -o pipefail
export THZ_BENCH_DEVICE=0 THZ_RCCL_LIB=$PWD/tests/mock_rccl/librccl_mock.so HSA_ENABLE_IPC_MODE_LEGACY=0
run() {  # ranks, then bench arguments
    n=$1; shift
    echo "== python -m torch.distributed.run --nproc-per-node $n bench.py --gpus $n $*"
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $n "$@" 2>&1 \
        | grep -v "^\[thzgpu\] THZ_RCCL_LIB\|amdgpu.ids\|OMP_NUM_THREADS\|^\*\*\*\*\*\|^W1\|^$"
}
run 4 --steps 5 --warmup 2 --no-cpu-baseline
run 2 --steps 5 --warmup 2 --no-cpu-baseline --wiener
run 2 --steps 3 --warmup 1 --no-cpu-baseline --nx 64 --ny 64 --nt 1024 --gather time
run 3 --steps 3 --warmup 1 --no-cpu-baseline --nx 64 --ny 64 --nt 1024 --gather all
