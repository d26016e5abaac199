#!/bin/bash
# Variant of libsvae_hip.so whose reverse-mode unit for latent dimension <n> is compiled with extra flags:
#   tools/build_vjp_variant.sh <out.so> <n> [extra hipcc flags...]
set -e
OUT=$(realpath -m "$1"); N=$2; shift 2
cd "$(dirname "$0")/../svae_amd/csrc"
TMP=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -DSVAE_N=$N -c lds_vjp_n.hip -o $TMP/v.o
OBJS=$(ls build/*.o | grep -v "lds_vjp_n$N.o" | grep -v sgb5)
mkdir -p "$(dirname "$OUT")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $TMP/v.o -o "$OUT"
rm -rf $TMP
