#!/bin/bash
# Round-4 closing run: full GPU suite, smoke, the default bench line
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO; mkdir -p gpurun_out
timeout 2400 python -m pytest -m gpu -q tests > gpurun_out/pytest_full.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" gpurun_out/pytest_full.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/pytest_full.log | head -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/bench_r4_final.json 2> gpurun_out/bench_r4_final.err; tail -c 300 gpurun_out/bench_r4_final.err
python - <<'PY'
import json
o=json.load(open('gpurun_out/bench_r4_final.json'))
print({k:o[k] for k in ('value','ms_per_step')}, o['roofline']['kernel_ms'], o['roofline']['frac'], o['roofline']['valu']['frac'], o['roofline']['traffic_over_algorithmic'])
for i,e in enumerate(o['extra']):
    print(i, {k:(round(v,4) if isinstance(v,float) else v) for k,v in e.items() if k not in('roofline','workload','kernel','ms_per_pass_all')}, (e.get('roofline') or {}).get('frac'))
c=o['cpu_baseline']; print(c['value'], c['cores'], c.get('core_info'), o.get('speedup_vs_cpu_baseline'))
PY
