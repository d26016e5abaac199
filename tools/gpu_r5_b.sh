#!/bin/bash
# Round 5, call B: bring-up of the producer-wavefront SLDS mean-field kernel (hang-safe: everything under `timeout`)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO; mkdir -p gpurun_out/r5b
timeout 240 python tools/slds_rpc_debug.py --only rpc_ref > gpurun_out/r5b/debug_ref.log 2>&1; echo "debug ref rc=$?"; tail -45 gpurun_out/r5b/debug_ref.log
timeout 240 python tools/slds_rpc_debug.py --only rpc_mfma --time > gpurun_out/r5b/debug_mfma.log 2>&1; echo "debug mfma rc=$?"; tail -60 gpurun_out/r5b/debug_mfma.log
timeout 600 python -m pytest tests/test_slds_hip.py -m gpu -q -x -k "fused_lds_meanfield_step" 2>&1 | tail -5
