#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest -m gpu -q tests/test_vjp_hip.py tests/test_svae_hip.py tests/test_slds_hip.py > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest.log
for B in 512 2304; do timeout 120 python tools/bench_train_path.py $B 200 10 1 | tail -1; done
