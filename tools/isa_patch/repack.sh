#!/bin/bash
# Re-assemble a hand-edited copy of a per-n kernel unit's gfx950 ISA into a variant of libsvae_hip.so (no recompilation of the
# C++): the way round 6 dumped registers from inside a loop that any source-level instrumentation "fixed".
#   tools/isa_patch/repack.sh prepare <unit> <n> [hipcc flags..]   -> work dir tools/isa_patch/_work/<unit>_n<n>/ with orig.s (device ISA)
#   tools/isa_patch/repack.sh build <unit> <n> <patched.s> <out.so>
# (run `make -C svae_amd/csrc` first: every other object comes from the regular build)
set -e
HERE=$(cd "$(dirname "$0")" && pwd); CSRC=$HERE/../../svae_amd/csrc
MODE=$1; UNIT=$2; N=$3; W=$HERE/_work/${UNIT}_n$N
if [ "$MODE" = prepare ]; then
  shift 3; mkdir -p "$W"; cd "$W"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSVAE_N=$N "$@" -c "$CSRC/$UNIT.hip" -o unit.o -save-temps -### 2>&1 | grep '^ "' > cmds.txt
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSVAE_N=$N "$@" -c "$CSRC/$UNIT.hip" -o unit.o -save-temps 2>/dev/null
  cp $UNIT-hip-amdgcn-amd-amdhsa-gfx950.s orig.s; echo "$W/orig.s"
else
  P=$(realpath "$4"); OUT=$(realpath -m "$5"); cd "$W"
  cp "$P" $UNIT-hip-amdgcn-amd-amdhsa-gfx950.s
  for i in 4 5 6 8 9 10; do eval "$(sed -n ${i}p cmds.txt)"; done      # device: as, lld, bundle; host: embed the new fat binary, as
  OBJS=$(ls $CSRC/build/*.o | grep -v "build/${UNIT}${N}.o" | grep -v sgb5 | grep -v ring2)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS unit.o -o "$OUT"
fi
