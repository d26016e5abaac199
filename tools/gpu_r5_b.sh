#!/bin/bash
# Round 5, call B: bring-up of the producer-wavefront SLDS mean-field kernel (hang-safe: everything under `timeout`)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO; mkdir -p gpurun_out/r5b
timeout 240 python tools/slds_rpc_debug.py --only rpc_ref > gpurun_out/r5b/debug_ref.log 2>&1; echo "debug ref rc=$?"; grep -E "MISMATCH|ALL OK|Error|error" gpurun_out/r5b/debug_ref.log | head
timeout 300 python tools/slds_rpc_debug.py --only rpc_mfma --time > gpurun_out/r5b/debug_mfma.log 2>&1; echo "debug mfma rc=$?"; grep -E "MISMATCH|ALL OK|Error|error|ms per" gpurun_out/r5b/debug_mfma.log | head -40
timeout 1200 python -m pytest tests/test_slds_hip.py tests/test_distributed_hip.py tests/test_models_hip.py tests/test_svae_hip.py tests/test_vjp_hip.py tests/test_lds_tile_hip.py -m gpu -q -s 2>&1 | grep -v "^\s*$" > gpurun_out/r5b/pytest.log; echo "pytest rc=${PIPESTATUS[0]}"; grep -E "passed|failed" gpurun_out/r5b/pytest.log | tail -3; grep -E "^FAILED|^ERROR" gpurun_out/r5b/pytest.log | head -20; grep -E "worst rel err" gpurun_out/r5b/pytest.log | head -20
timeout 300 python - <<'PY'
import json, torch, bench
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
print(json.dumps(bench.measure_slds(dev)))
PY
