#!/bin/bash
# Round 5, call C: a changed SLDS producer kernel -- parity vs the table kernel, timing, SLDS tests, ascent
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO; mkdir -p gpurun_out/r5c
timeout 300 python tools/slds_rpc_debug.py --sweep 2>&1 | grep -v amdgpu | grep -c "ok  "; timeout 300 python tools/slds_rpc_debug.py --sweep 2>&1 | grep MISMATCH | head
timeout 300 python tools/slds_rpc_debug.py --only default --time > gpurun_out/r5c/debug_default.log 2>&1; echo "debug default rc=$?"; grep -E "MISMATCH|ALL OK|Error|error|ms per|off" gpurun_out/r5c/debug_default.log | head -60
timeout 900 python -m pytest tests/test_slds_hip.py tests/test_distributed_hip.py -m gpu -q 2>&1 | tail -4
timeout 300 python - <<'PY'
import json, torch, bench
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
print(json.dumps(bench.measure_slds(dev)))
PY
