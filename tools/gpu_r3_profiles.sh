#!/bin/bash
# Round-3 rocprofv3 evidence, one gpurun call: bash tools/gpu_r3_profiles.sh  (summaries: profiles/summarize*.py)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
bash profiles/run_profile.sh r3_twoend 2>&1 | tail -2
bash profiles/run_profile.sh r3_twoend_b4096 --seqs-per-gpu 4096 2>&1 | tail -2
ARGS_TILE="--workload lds64"
OUT=$REPO/gpurun_out/prof_r3_tile_n64_b512; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp
  A="--steps 4 --warmup 1 --no-cpu-baseline --no-extra --workload lds64"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $REPO/bench.py $A > $OUT/trace.log 2>&1
  timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python $REPO/bench.py $A > $OUT/pmc_fetch.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python $REPO/bench.py $A > $OUT/pmc_write.log 2>&1
  find $OUT -name "*kernel_trace.csv" -size +2M -delete )
bash tools/prof_train.sh r3_train 512 200 10 1 2>&1 | tail -3
bash tools/prof_train.sh r3_train_b4096 4096 200 10 1 2>&1 | tail -3
python bench.py > gpurun_out/bench_r3b.json 2> gpurun_out/bench_r3b.err; tail -c 600 gpurun_out/bench_r3b.json
