#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO; mkdir -p gpurun_out
timeout 1200 python -m pytest -m gpu -q -x tests/test_hmm_hip.py tests/test_slds_hip.py > gpurun_out/pytest_slds.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/pytest_slds.log
timeout 300 python tools/bench_slds.py 2048 500 10 8 --fused-only 2>&1 | grep -v amdgpu | tail -6
