#!/bin/bash
# two-ended kernel experiments: LDS parity suite + headline timing
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO; mkdir -p gpurun_out
timeout 1200 python -m pytest -m gpu -q -x tests/test_lds_hip.py > gpurun_out/pytest_te.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/pytest_te.log
for i in 1 2 3; do python bench.py --no-extra --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; o=json.loads(sys.stdin.read()); print('headline', o['value'], o['ms_per_step'], o['roofline']['kernel_ms'])"; done
python bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; o=json.loads(sys.stdin.read()); print('driver-like', o['value'], o['ms_per_step'], o['roofline']['kernel_ms'])"
python bench.py --no-extra --no-cpu-baseline --steps 50 --warmup 5 --seqs-per-gpu 256 2>/dev/null | python -c "import json,sys; o=json.loads(sys.stdin.read()); print('B=256', o['value'], o['ms_per_step'], o['roofline']['kernel_ms'])"
