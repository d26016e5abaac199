#!/bin/bash
# Round-4 closing run: profiles of the kernels that changed after parts A / B, then tools/gpu_r4_final.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
bash profiles/run_profile.sh r4_twoend 2>&1 | tail -1
( cd /tmp && export TMPDIR=/tmp; OUT=$REPO/gpurun_out/prof_r4_twoend
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -o bench -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra > $OUT/pmc_sq.log 2>&1 )
bash tools/prof_generic.sh r4_train python $REPO/tools/bench_train_path.py 512 200 10 1 2>&1 | tail -1
bash tools/prof_generic.sh r4_gmm python $REPO/tools/bench_gmm_step.py 2>&1 | tail -1
bash tools/gpu_r4_final.sh
