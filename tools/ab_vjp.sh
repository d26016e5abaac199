#!/bin/bash
# A/B of library variants on the training path (same box, alternating): tools/ab_vjp.sh <reps> <variant.so> [variant2.so ...]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
R=$1; shift
for rep in $(seq $R); do
  unset SVAE_AMD_LIB; echo -n "default: "; python tools/bench_train_path.py 512 200 10 1 2>&1 | grep "training path" | cut -c27-150
  for V in "$@"; do
    export SVAE_AMD_LIB=$REPO/$V; echo -n "$(basename $V): "; python tools/bench_train_path.py 512 200 10 1 2>&1 | grep "training path" | cut -c27-150
  done
done
