#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (tools/ubench/fetch_calibration.hip):  bash tools/gpu_fetch_calibration.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/fetch_cal; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp
  timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o cal -- $REPO/variants/fetch_cal > $OUT/fetch.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o cal -- $REPO/variants/fetch_cal > $OUT/write.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o cal -- $REPO/variants/fetch_cal > $OUT/trace.log 2>&1 )
cd $REPO; python tools/ubench/fetch_calibration.py $OUT $OUT/calibration.json
cut -d, -f1-4 $OUT/trace/*/cal_kernel_stats.csv 2>/dev/null | head -12 || find $OUT/trace -name "*stats*"
