#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest -m gpu -q -x tests/test_gmm_hip.py tests/test_models_hip.py tests/test_svae_hip.py > gpurun_out/pytest_gmm.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/pytest_gmm.log
timeout 600 python tools/bench_gmm.py 2>&1 | grep -v amdgpu.ids
