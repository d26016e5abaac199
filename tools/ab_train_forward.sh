#!/bin/bash
# A/B of the training path's forward pass (two-ended CROSS kernel || one-directional filter) between library variants
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for rep in 1 2; do
for v in default n10_filter_shuffle n10_cross_shuffle n10_both_shuffle; do
  if [ $v = default ]; then unset SVAE_AMD_LIB; else export SVAE_AMD_LIB=$REPO/tests/_variants/$v.so; fi
  echo -n "$v: "; python tools/bench_train_path.py 512 200 10 1 2>&1 | grep "training path" | cut -c1-150
done; done
