#!/bin/bash
# Build an experimental variant of libsvae_hip.so whose E-step unit for latent dimension <n> is compiled
# with extra flags; every other object comes from the regular build (run `make -C svae_amd/csrc` first):
#   tools/build_variant.sh <out.so> <n> [extra hipcc -D flags...]
set -e
OUT=$(realpath -m "$1"); N=$2; shift 2
cd "$(dirname "$0")/../svae_amd/csrc"
TMP=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -DSVAE_N=$N -c lds_estep_n.hip -o $TMP/n.o
OBJS=$(ls build/*.o | grep -v "lds_estep_n$N.o" | grep -v sgb5)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $TMP/n.o -o "$OUT"
rm -rf $TMP
