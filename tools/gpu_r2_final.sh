#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest -m gpu -q tests > gpurun_out/pytest_full.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/pytest_full.log
