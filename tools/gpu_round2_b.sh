#!/bin/bash
# LDS tests through every kernel, A/B bench, phase timing, rocprofv3 evidence for the two-ended kernel
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
TE_MODE=1 timeout 300 python tools/dbg_twoend.py > gpurun_out/dbg_twoend1.log 2>&1; tail -1 gpurun_out/dbg_twoend1.log
timeout 900 python -m pytest tests/test_lds_hip.py -m gpu -q > gpurun_out/t_lds.log 2>&1; echo "pytest lds rc=$?"; tail -3 gpurun_out/t_lds.log
for te in twoend twoend_full; do for B in 512 1024 4096; do
  timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --kernel $te --seqs-per-gpu $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('kernel=$te B=$B %.1f us kernel, %.0f seq/s' % (1e3*d['roofline']['kernel_ms'], d['value']))"
done; done
SVAE_AMD_LIB=$PWD/variants/te_timing.so timeout 120 python tools/te_phase_timing.py 2>&1 | tail -3
bash profiles/run_profile.sh r2_twoend 2>&1 | tail -2
bash profiles/run_profile.sh r2_twoend_b4096 --seqs-per-gpu 4096 2>&1 | tail -1
