#!/bin/bash
# GMM kernels: parity suite (all three forms) + the two-rank protocol test + timing
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO; mkdir -p gpurun_out
timeout 900 python -m pytest -m gpu -q -x tests/test_gmm_hip.py tests/test_models_hip.py tests/test_svae_hip.py "tests/test_distributed_hip.py" tests/test_abi.py > gpurun_out/pytest_gmm.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/pytest_gmm.log
timeout 300 python tools/bench_gmm.py 2>&1 | tee gpurun_out/bench_gmm.log
