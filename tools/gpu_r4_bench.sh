#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO; mkdir -p gpurun_out
python bench.py > gpurun_out/bench_r4b.json 2> gpurun_out/bench_r4b.err; tail -c 300 gpurun_out/bench_r4b.err
python - <<'PY'
import json
o=json.load(open('gpurun_out/bench_r4b.json'))
print({k:o[k] for k in ('value','ms_per_step')}, o['roofline']['kernel_ms'], o['roofline']['frac'])
for i,e in enumerate(o['extra']):
    print(i, {k:(round(v,4) if isinstance(v,float) else v) for k,v in e.items() if k not in('roofline','workload','kernel','ms_per_pass_all')})
c=o['cpu_baseline']; print(c['value'], c['cores'], c.get('core_info'))
PY
