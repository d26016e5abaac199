#!/bin/bash
# rocprofv3 kernel-trace summaries (+ HBM counters for the SLDS kernels) of the non-headline kernels:
#   gpurun_out/prof_r2_slds (SLDS ascent + run_inference at BASELINE configs[3]), gpurun_out/prof_r2_gmm
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$REPO/gpurun_out/prof_r2_slds; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $REPO/tools/bench_slds.py 2048 500 10 8 --fused-only --run-inference > $O/bench.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- python $REPO/tools/bench_slds.py 2048 500 10 8 --fused-only > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- python $REPO/tools/bench_slds.py 2048 500 10 8 --fused-only > $O/pmc_write.log 2>&1
grep -v "^[EW]2026" $O/bench.log | tail -5
O=$REPO/gpurun_out/prof_r2_gmm; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $REPO/tools/bench_gmm.py > $O/bench.log 2>&1
grep -v "^[EW]2026" $O/bench.log | tail -8
find $REPO/gpurun_out -name "*kernel_trace.csv" -size +2M -delete
