#!/bin/bash
# builds sim_bench[_<tag>] with optional extra -D flags:  ./build.sh tag -DABL_X
cd "$(dirname "$0")"
TAG=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 "$@" -I../../include sim_bench.cpp ../../protein-clip_amd/csrc/filip_stream.hip ../../protein-clip_amd/csrc/filip_fast.hip -o sim_bench${TAG:+_$TAG}
