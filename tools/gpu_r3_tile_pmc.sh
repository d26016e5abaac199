#!/bin/bash
# rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes of the tile E-step at BASELINE configs[4] shape (512 sequences),
# then the default bench line:  bash tools/gpu_r3_tile_pmc.sh   (summaries: python profiles/summarize.py r3_tile_n64_b512 lds_estep_tile)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_r3_tile_n64_b512; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp
  A="--steps 4 --warmup 1 --no-cpu-baseline --no-extra --workload lds64"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $REPO/bench.py $A > $OUT/trace.log 2>&1
  timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python $REPO/bench.py $A > $OUT/pmc_fetch.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python $REPO/bench.py $A > $OUT/pmc_write.log 2>&1
  find $OUT -name "*kernel_trace.csv" -size +2M -delete )
head -4 $OUT/trace/bench_kernel_stats.csv | cut -c1-160
cd $REPO
python profiles/summarize.py r3_tile_n64_b512 lds_estep_tile > /dev/null 2>&1; cp profiles/r3_tile_n64_b512/pmc_hbm.json gpurun_out/r3_tile_pmc_hbm.json
python bench.py > gpurun_out/bench_r3_final.json 2> gpurun_out/bench_r3_final.err; tail -c 300 gpurun_out/bench_r3_final.json
