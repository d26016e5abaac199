#!/bin/bash
# Build an experimental variant of libsvae_hip.so in which ONE unit (e.g. gmm_meanfield) is compiled with extra flags;
# every other object comes from the regular build (run `make -C svae_amd/csrc` first):
#   tools/build_unit_variant.sh <out.so> <unit> [extra hipcc flags...]
set -e
OUT=$(realpath -m "$1"); UNIT=$2; shift 2
cd "$(dirname "$0")/../svae_amd/csrc"
TMP=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $UNIT.hip -o $TMP/u.o
OBJS=$(ls build/*.o | grep -v "build/$UNIT.o" | grep -v sgb5)
mkdir -p "$(dirname "$OUT")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $TMP/u.o -o "$OUT"
rm -rf $TMP
