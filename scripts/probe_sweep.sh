export THZ_ONLY=probe
for cfg in "256 512 0" "256 512 1" "512 512 0" "1024 256 0" "2048 256 0" "4096 256 0" "2048 256 1" "512 256 0" "8192 64 0"; do
  set -- $cfg
  echo "blocks=$1 threads=$2 nt_store=$3"
  THZ_PROBE_BLOCKS=$1 THZ_PROBE_THREADS=$2 THZ_PROBE_NT=$3 timeout -k 10 100 python scripts/gpu_pipe_timing.py 512 1024 4096 || exit 1
done
