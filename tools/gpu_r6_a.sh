#!/bin/bash
# round 6, first pass on the box: lean training path -- tests that cover it, profiles at 4096 (trace + counters)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
python -m pytest tests/test_lean_hip.py tests/test_vjp_hip.py tests/test_lds_hip.py tests/test_models_hip.py tests/test_svae_hip.py -x -q 2>&1 | tail -8
bash tools/prof_generic.sh r6_train_b4096 python $REPO/tools/bench_train_path.py 4096 200 10 1 2>&1 | tail -3
cd $REPO && python profiles/summarize_all.py r6_train_b4096 lean 2>&1 | tail -40
