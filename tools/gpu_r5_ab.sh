#!/bin/bash
# same-box A/B of two builds of the SLDS kernels: tests/_variants/exp_old.so vs the in-tree library, alternating
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for rep in 1 2; do
  for v in old new; do
    if [ $v = old ]; then export SVAE_AMD_LIB=$REPO/tests/_variants/exp_old.so; else unset SVAE_AMD_LIB; fi
    echo "== $v (rep $rep)"; timeout 200 python tools/slds_rpc_debug.py --only default --time 2>&1 | grep -E "(default|rpc_mfma) +[0-9.]+ ms|MISMATCH" | grep -E "B= +(8|256|512|2048) "
  done
done
