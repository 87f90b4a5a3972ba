#!/bin/bash
# Builds a variant of libpclip.so into build_ab/<name>.so: tools/build_variant.sh <name> [extra hipcc flags for filip_stream.hip ...]
# (only filip_stream.hip is rebuilt with the extra flags; the other objects come from the normal build)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $ROOT/build_ab
make -C $ROOT/protein-clip_amd/csrc -j8 > /dev/null
cd $ROOT/protein-clip_amd/csrc
SRC=${VARIANT_SRC:-filip_stream.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include "$@" -c $SRC -o $ROOT/build_ab/$NAME.o
OBJS=$(ls *.o | grep -v "^${VARIANT_REPLACES:-filip_stream}.o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $ROOT/build_ab/$NAME.o -o $ROOT/build_ab/$NAME.so
echo built build_ab/$NAME.so
