#!/bin/bash
# A/B of library variants on the training path (same box, alternating): tools/ab_vjp.sh <variant.so> [reps]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
V=$1; R=${2:-3}
for rep in $(seq $R); do
  unset SVAE_AMD_LIB; echo -n "default: "; python tools/bench_train_path.py 512 200 10 1 2>&1 | grep "training path" | cut -c1-140
  export SVAE_AMD_LIB=$REPO/$V; echo -n "variant: "; python tools/bench_train_path.py 512 200 10 1 2>&1 | grep "training path" | cut -c1-140
done
