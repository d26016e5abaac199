#!/bin/bash
# first GPU call of round 2: two-ended kernel vs packed, LDS test file, three-kernel bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for m in 1 2; do
TE_MODE=$m timeout 300 python tools/dbg_twoend.py > gpurun_out/dbg_twoend$m.log 2>&1; echo "dbg mode $m rc=$?"
tail -2 gpurun_out/dbg_twoend$m.log
grep -c "<--" gpurun_out/dbg_twoend$m.log
done
timeout 900 python -m pytest tests/test_lds_hip.py -m gpu -q -k "twoend" > gpurun_out/t_lds_twoend.log 2>&1; echo "pytest twoend rc=$?"
tail -5 gpurun_out/t_lds_twoend.log
for te in twoend twoend_full packed; do for B in 512 1024 4096; do
  timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --kernel $te --seqs-per-gpu $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('kernel=$te B=$B %.1f us kernel, %.0f seq/s' % (1e3*d['roofline']['kernel_ms'], d['value']))"
done; done
SVAE_AMD_LIB=$PWD/variants/te_timing.so timeout 120 python tools/te_phase_timing.py 2>&1 | tail -4
