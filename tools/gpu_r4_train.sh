#!/bin/bash
# training path (n = 10): sampler / VJP / filter parity + timings
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO; mkdir -p gpurun_out
timeout 1500 python -m pytest -m gpu -q -x tests/test_lds_hip.py tests/test_vjp_hip.py tests/test_svae_hip.py tests/test_models_hip.py > gpurun_out/pytest_train.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/pytest_train.log
python tools/bench_train_path.py 512 200 10 1 2>&1 | grep -v amdgpu | tail -2
python tools/bench_train_path.py 4096 200 10 1 2>&1 | grep -v amdgpu | tail -2
