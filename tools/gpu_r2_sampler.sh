#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1200 python -m pytest -m gpu -q -x tests/test_lds_hip.py tests/test_vjp_hip.py tests/test_slds_hip.py tests/test_svae_hip.py tests/test_models_hip.py > gpurun_out/pytest_sampler.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/pytest_sampler.log
timeout 300 python tools/bench_train_path.py 512 200 10 1 2>&1 | tail -2
timeout 300 python tools/bench_train_path.py 512 200 10 4 2>&1 | tail -2 | head -1
timeout 300 python tools/bench_train_path.py 2304 200 10 1 2>&1 | tail -2 | head -1
