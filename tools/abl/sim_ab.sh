#!/bin/bash
R=$GRAFT_REPO_ROOT
cp $R/protein-clip_amd/libpclip.so /tmp/lib_orig.so
for v in "$@"; do
  cp $R/tools/abl/lib$v.so $R/protein-clip_amd/libpclip.so
  echo "== $v"
  timeout -k 10 100 python3 $R/tools/sim_digest.py 2>&1 | tail -4 | cut -c1-100 || break
  timeout -k 10 200 python3 $R/tools/sweep_stream_wgs.py ragged 768 768 2>&1 | grep round || break
  timeout -k 10 200 python3 $R/tools/sweep_stream_wgs.py full 768 2>&1 | grep round | tail -1 || break
done
cp /tmp/lib_orig.so $R/protein-clip_amd/libpclip.so
