#!/bin/bash
# fused SLDS mean field: parity tests, timing at BASELINE configs[3], rocprofv3 summary
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest -m gpu -q -x tests/test_slds_hip.py tests/test_vjp_hip.py tests/test_abi.py > gpurun_out/pytest_slds.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/pytest_slds.log
timeout 600 python tools/bench_slds.py 2048 500 10 8 --run-inference 2>&1 | tail -8
timeout 600 python tools/slds_sections.py 2>&1 | grep -v amdgpu.ids | tail -30
SVAE_AMD_LIB=$PWD/variants/te_timing.so timeout 300 python tools/te_mix_phase_timing.py 2>&1 | tail -1
SVAE_AMD_LIB=$PWD/variants/te_timing.so timeout 300 python tools/te_mix_phase_timing.py 256 500 8 2>&1 | tail -1
