#!/bin/bash
# A/B of library builds on one GPU box for the FILIP score kernel: result digests (tools/sim_digest.py) and kernel times
# (tools/sim_time.sh, ragged and full-length benchmark shapes) of every build listed, the builds taking turns.
#   usage: tools/sim_ab.sh <rounds> libA.so [libB.so ...]       (paths relative to the repository root)
set -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
ROUNDS=$1; shift
for lib in "$@"; do
    cp $ROOT/$lib $ROOT/protein-clip_amd/libpclip.so || exit 1
    echo "== $(basename $lib .so) digest: $(timeout -k 10 120 python3 $ROOT/tools/sim_digest.py 2>&1 | tr '\n' ' ' | cut -c1-400)"
done
for r in $(seq 1 $ROUNDS); do
    for lib in "$@"; do
        cp $ROOT/$lib $ROOT/protein-clip_amd/libpclip.so || exit 1
        echo "== $(basename $lib .so) round $r"
        bash $ROOT/tools/sim_time.sh ab_$(basename $lib .so)_$r || exit 1
    done
done
