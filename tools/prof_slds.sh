#!/bin/bash
# rocprofv3 kernel trace of the SLDS local mean field + run_inference at BASELINE configs[3]
TAG=${1:-r2_slds}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $REPO/tools/bench_slds.py 2048 500 10 8 --fused-only --run-inference "$@" > $OUT/bench.log 2>&1
tail -6 $OUT/bench.log
find $OUT -name "*kernel_trace.csv" -size +2M -delete
head -25 $OUT/trace/bench_kernel_stats.csv | cut -c1-160
