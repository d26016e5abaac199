#!/bin/bash
# Round-4 rocprofv3 evidence, part A: the headline kernel (512 and 4096 sequences) and the GMM kernels
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
bash profiles/run_profile.sh r4_twoend 2>&1 | tail -2
bash profiles/run_profile.sh r4_twoend_b4096 --seqs-per-gpu 4096 2>&1 | tail -2
bash tools/prof_generic.sh r4_gmm python $REPO/tools/bench_gmm_step.py 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_r4_twoend
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -o bench -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra > $OUT/pmc_sq.log 2>&1
