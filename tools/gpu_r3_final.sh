#!/bin/bash
# Round-3 closing run, one gpurun call: full GPU suite, the default bench line (-> profiles/r3_bench), SLDS trace.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
bash tools/gpu_run_tests.sh tests
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/bench_r3_final.json 2> gpurun_out/bench_r3_final.err; tail -c 400 gpurun_out/bench_r3_final.json
bash tools/prof_slds.sh r3_slds 2>&1 | tail -30
