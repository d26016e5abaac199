#!/bin/bash
# On the GPU box: parity check + bench of every variants/*.so (n = 10 only).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for so in variants/*.so; do
  for rows in 4 1; do
    export SVAE_AMD_LIB=$PWD/$so SVAE_LDS_ROWS_PER_WAVE=$rows
    ok=$(timeout 120 python -m pytest tests/test_lds_hip.py -m gpu -q -k "golden and (T200 or T20_n10)" 2>&1 | tail -1)
    for B in 512 4096; do
      r=$(timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --seqs-per-gpu $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.1f us kernel, %.0f seq/s' % (1e3*d['roofline']['kernel_ms'], d['value']))")
      echo "$so rows=$rows B=$B: $r   [$ok]"
    done
  done
done
