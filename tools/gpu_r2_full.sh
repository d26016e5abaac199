#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest -m gpu -q -x tests > gpurun_out/pytest_full.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/pytest_full.log
timeout 300 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench %.0f seq/s, kernel %.1f us' % (d['value'], 1e3*d['roofline']['kernel_ms']))"
SVAE_AMD_LIB=$PWD/variants/te_timing.so timeout 120 python tools/te_phase_timing.py 2>&1 | tail -3
