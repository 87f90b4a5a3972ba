#!/bin/bash
# On the GPU box: FILIP backward tests, then a kernel trace of the benchmark (FILIP only).
#   usage: bash tools/gpu_check_bwd.sh <tag>
set -o pipefail
TAG=${1:-x}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_kernels.py -m gpu -x -q --timeout 200 --timeout-method thread -k "filip and (backward or routing or ties or fused or random_small)" > "$OUT/pytest.log" 2>&1 || { tail -40 "$OUT/pytest.log"; exit 1; }
tail -2 "$OUT/pytest.log"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --workload filip > "$OUT/trace.log" 2>&1 || { tail -30 "$OUT/trace.log"; exit 1; }
f=$(ls $OUT/trace/*/*_kernel_stats.csv | tail -1)
cut -d, -f1-4 "$f" | head -8
grep '^{' "$OUT/trace.log" | cut -c1-300
