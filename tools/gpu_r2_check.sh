#!/bin/bash
# round-2 state check: whole GPU suite, default bench line, training-path + bench rocprofv3 evidence
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest -m gpu -q -x --durations=15 tests > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" gpurun_out/pytest.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/pytest.log | head -20
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cat gpurun_out/bench.json
for B in 512 2304; do timeout 120 python tools/bench_train_path.py $B 200 10 1 | tail -1; done
bash tools/prof_train.sh r2_train 512 200 10 1 2>&1 | tail -12
bash profiles/run_profile.sh r2_bench 2>&1 | tail -3
