#!/bin/bash
# rocprofv3 kernel trace of the n = 64 training pass:  bash tools/prof_tile_train.sh [B [tag]]   (default 64 sequences -> gpurun_out/prof_r3_tile_train)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-64}; TAG=${2:-r3_tile_train}
O=$REPO/gpurun_out/prof_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $REPO/tools/bench_tile_train.py $B 1000 64 1 > $O/bench.log 2>&1
grep -v "^[EW]2026" $O/bench.log | tail -2
find $O -name "*kernel_trace.csv" -size +2M -delete
head -14 $O/trace/bench_kernel_stats.csv | cut -c1-140
