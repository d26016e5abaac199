#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest -m gpu -q -x tests/test_lds_tile_hip.py > gpurun_out/pytest_tile.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/pytest_tile.log
timeout 400 python tools/bench_tile_train.py 64 1000 64 1 2>&1 | tail -1
timeout 400 python tools/bench_tile_train.py 512 200 32 1 2>&1 | tail -1
