#!/bin/bash
# Build a variant of libsvae_hip.so whose tiled-path unit is compiled with extra flags:
#   tools/build_tile_variant.sh <out.so> [extra hipcc flags...]      (run `make -C svae_amd/csrc` first)
set -e
OUT=$(realpath -m "$1"); shift
cd "$(dirname "$0")/../svae_amd/csrc"
TMP=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c lds_estep_tile.hip -o $TMP/tile.o
OBJS=$(ls build/*.o | grep -v "lds_estep_tile" )
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $TMP/tile.o -o "$OUT"
rm -rf $TMP
