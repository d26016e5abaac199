#!/bin/bash
# Round-3 closing run, one gpurun call: full GPU suite, smoke, the rocprofv3 evidence of the kernels that changed since
# tools/gpu_r3_profiles.sh (tile E-step counters, training path, SLDS, tile training), then the default bench line.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
bash tools/gpu_run_tests.sh tests
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
OUT=$REPO/gpurun_out/prof_r3_tile_n64_b512; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp
  A="--steps 4 --warmup 1 --no-cpu-baseline --no-extra --workload lds64"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $REPO/bench.py $A > $OUT/trace.log 2>&1
  timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python $REPO/bench.py $A > $OUT/pmc_fetch.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python $REPO/bench.py $A > $OUT/pmc_write.log 2>&1
  find $OUT -name "*kernel_trace.csv" -size +2M -delete )
if [ -z "$TILE_ONLY" ]; then     # (TILE_ONLY=1: only the tile kernels changed since the last full run)
bash tools/prof_train.sh r3_train 512 200 10 1 2>&1 | tail -3
bash tools/prof_train.sh r3_train_b4096 4096 200 10 1 2>&1 | tail -3
bash tools/prof_slds.sh r3_slds 2>&1 | grep -v "^[EW]2026" | tail -8 | cut -c1-160
fi
bash tools/prof_tile_train.sh 2>&1 | head -3 | cut -c1-160
cd $REPO
python bench.py > gpurun_out/bench_r3_final.json 2> gpurun_out/bench_r3_final.err; tail -c 300 gpurun_out/bench_r3_final.json
