#!/bin/bash
# Kernel time of the FILIP score kernel alone (rocprofv3 kernel trace over tools/run_filip_fwd.py), ragged and
# full-length benchmark shapes; prints the average of the launches after the first.   usage: tools/sim_time.sh [tag]
set -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/simtime_${1:-x}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for shape in ragged full; do
    extra=""; [ $shape = full ] && extra="full"
    timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $OUT/$shape -o t -- python3 $ROOT/tools/run_filip_fwd.py 80 unit $extra > $OUT/$shape.log 2>&1 || { tail -5 $OUT/$shape.log; exit 1; }
    python3 - $OUT/$shape/t_kernel_trace.csv $shape <<'PY'
import csv, sys
d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(sys.argv[1])) if r['Kernel_Name'].startswith('filip_sim_stream')]
d = d[40:]
print(f"{sys.argv[2]}: filip_sim_stream_kernel {sum(d) / len(d) / 1e3:.1f} us average of {len(d)} (min {min(d) / 1e3:.1f})")
PY
done
