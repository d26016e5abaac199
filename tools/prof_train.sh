#!/bin/bash
# rocprofv3 evidence for the training path (E-step keeping the hand-off, sampler, VJP sweeps) at the
# headline shape: tools/prof_train.sh <tag> [B T n S]   ->  gpurun_out/prof_<tag>/
TAG=${1:-r2_train}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 python $REPO/tools/bench_train_path.py "$@" > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $REPO/tools/bench_train_path.py "$@" > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python $REPO/tools/bench_train_path.py "$@" > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python $REPO/tools/bench_train_path.py "$@" > $OUT/pmc_write.log 2>&1
find $OUT -name "*kernel_trace.csv" -size +2M -delete
head -8 $OUT/trace/bench_kernel_stats.csv | cut -c1-150
