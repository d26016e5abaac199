#!/bin/bash
# usage: tools/gpu_run_tests.sh <pytest args...>   (log in gpurun_out/pytest.log)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest -m gpu -q "$@" > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" gpurun_out/pytest.log | tail -5
grep -E "^FAILED|^ERROR" gpurun_out/pytest.log | head -20
