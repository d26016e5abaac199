#!/bin/bash
# Build an experimental single-latent-dimension variant of libsvae_hip.so:
#   tools/build_variant.sh <out.so> <n> [extra hipcc -D flags...]
set -e
OUT=$1; N=$2; shift 2
cd "$(dirname "$0")/../svae_amd/csrc"
TMP=$(mktemp -d)
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC $*"
/opt/rocm/bin/hipcc $F -DSVAE_N=$N -c lds_estep_n.hip -o $TMP/n.o &
/opt/rocm/bin/hipcc $F -DSVAE_ONLY_N=$N -c lds_estep.hip -o $TMP/d.o &
/opt/rocm/bin/hipcc $F -c gmm_meanfield.hip -o $TMP/g.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $TMP/n.o $TMP/d.o $TMP/g.o -o $OUT
rm -rf $TMP
