#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest -m gpu -q -x tests/test_models_hip.py tests/test_svae_hip.py tests/test_abi.py > gpurun_out/pytest_models.log 2>&1; echo "pytest rc=$?"
tail -12 gpurun_out/pytest_models.log
timeout 300 python tools/bench_train_path.py 512 200 10 1 2>&1 | tail -2
