#!/bin/bash
# A/B of library builds on one GPU box: per-kernel average durations of the FILIP step at the benchmark shape
# (rocprofv3 kernel trace over tools/run_filip_step.py), the builds taking turns.
#   usage: tools/step_time.sh <tag> <rounds> libA.so [libB.so ...]      (paths relative to the repository root)
# Every build is copied over protein-clip_amd/libpclip.so in turn; the last one listed stays in place.
set -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; ROUNDS=$2; shift 2
OUT=$ROOT/gpurun_out/steptime_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for r in $(seq 1 $ROUNDS); do
    for lib in "$@"; do
        name=$(basename $lib .so)
        cp $ROOT/$lib $ROOT/protein-clip_amd/libpclip.so || exit 1
        timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $OUT/${name}_$r -o t -- python3 $ROOT/tools/run_filip_step.py ${STEPS:-60} > $OUT/${name}_$r.log 2>&1 || { tail -5 $OUT/${name}_$r.log; exit 1; }
        python3 - $OUT/${name}_$r/t_kernel_trace.csv "$name round $r" <<'PY'
import collections, csv, sys
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r['Kernel_Name'].split('(')[0]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
out = []
tot = 0.0
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    per = len(v) // 60 if len(v) >= 60 else 1          # launches per step
    v = v[len(v) // 3:]                                # the first third is clock ramp
    avg = sum(v) / len(v) / 1e3
    tot += avg * per
    if avg * per > 3.0:
        out.append(f"{k.replace('filip_', '').replace('_kernel', '')} {per}x{avg:.1f}")
print(f"{sys.argv[2]}: step {tot:.1f} us | " + ", ".join(out))
PY
    done
done
