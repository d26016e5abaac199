#!/bin/bash
# Round 5, call A: the whole GPU suite (exhaustive full-size parity) + the default bench line with its parity gate
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO; mkdir -p gpurun_out/r5a
timeout 1500 python -m pytest tests -m gpu -q -x -s 2>&1 | grep -v "^\s*$" > gpurun_out/r5a/pytest.log; echo "pytest rc=${PIPESTATUS[0]}"
grep -E "passed|failed|error" gpurun_out/r5a/pytest.log | tail -3
grep -E "worst rel err|vs the compiled|vs the restatement" gpurun_out/r5a/pytest.log | sort | uniq | head -60
timeout 900 python bench.py > gpurun_out/r5a/bench.json 2> gpurun_out/r5a/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5a/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "parity", d.get("parity"))
print("roofline", {k: d["roofline"][k] for k in ("frac", "kernel_ms", "traffic", "traffic_source")})
for i, e in enumerate(d.get("extra", [])):
    print(i, {k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk in ("max_rel", "ok", "frac", "kernel_ms")}) for k, v in e.items() if k not in ("workload",)})
print("cpu", {k: d["cpu_baseline"].get(k) for k in ("value", "cores", "kind")})
PY
