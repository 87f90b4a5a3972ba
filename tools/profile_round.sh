#!/bin/bash
# Runs on the GPU box (through gpurun): GPU tests, the benchmark, a rocprofv3 kernel-trace of
# the same benchmark command, and the HBM-traffic counter passes of the dominant kernel
# (one --pmc pass per run, no tracing domains beside it).  Everything lands under
# gpurun_out/<tag>/; tools/summarise_profiles.py turns it into the files kept in profiles/.
#   usage: bash tools/profile_round.sh <tag> [steps]
set -o pipefail
TAG=${1:-r1}
STEPS=${2:-100}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp

if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > "$OUT/pytest_gpu.log" 2>&1 || { tail -30 "$OUT/pytest_gpu.log"; exit 1; }
tail -3 "$OUT/pytest_gpu.log"
fi

timeout -k 10 600 python3 bench.py --steps "$STEPS" --warmup 10 --full-length > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -30 "$OUT/bench.err"; exit 1; }
cat "$OUT/bench.json"

timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps "$STEPS" --warmup 10 --no-cpu-baseline --no-rank-shapes > "$OUT/trace.log" 2>&1 || { tail -30 "$OUT/trace.log"; exit 1; }

for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE"; do
    name=$(echo "$pass" | tr ' ' '_' | cut -c1-40)
    timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$name" -- python3 tools/run_filip_step.py 3 > "$OUT/pmc_$name.log" 2>&1 || { tail -20 "$OUT/pmc_$name.log"; exit 1; }
done
echo "profile_round $TAG done"
