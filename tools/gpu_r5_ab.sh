#!/bin/bash
# same-box A/B of two builds of the SLDS producer kernel: tests/_variants/exp_old.so vs the in-tree library, alternating
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for rep in 1 2; do
  for v in old prio3; do
    if [ $v = old ]; then export SVAE_AMD_LIB=$REPO/tests/_variants/exp_old.so; else export SVAE_AMD_LIB=$REPO/tests/_variants/exp_prio3.so; fi
    echo "== $v (rep $rep)"; timeout 200 python tools/slds_rpc_debug.py --only rpc_mfma --time 2>&1 | grep -E "rpc_mfma +[0-9.]+ ms|MISMATCH" | grep -E "B= +(8|512|1024|2048) "
  done
done
