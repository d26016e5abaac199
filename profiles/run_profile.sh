#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun):
#   1. --kernel-trace --stats           -> per-kernel average duration
#   2. --pmc FETCH_SIZE  (own pass)     -> HBM read bytes   (gfx950: x2 correction for wide coalesced
#   3. --pmc WRITE_SIZE  (own pass)        streams, MI355X_MICROARCH.md section HBM)
# Usage: profiles/run_profile.sh <tag> [bench args...]; writes gpurun_out/prof_<tag>/
set -u
TAG=${1:-r1}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
(cd $REPO && python -c "from svae_amd import _lib; print(_lib.source_hash())") > $OUT/csrc_sha16.txt   # what was profiled
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-extra $*"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $REPO/bench.py $ARGS > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python $REPO/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python $REPO/bench.py $ARGS > $OUT/pmc_write.log 2>&1
# 4. --pmc SQ_INSTS_VALU (own pass)  -> VALU instructions issued per launch (bench.py: roofline.valu.issued_over_algorithmic)
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d $OUT/pmc_valu -o bench -- python $REPO/bench.py $ARGS > $OUT/pmc_valu.log 2>&1
find $OUT -name "*.csv" | head -20
# keep only the small summaries
find $OUT -name "*kernel_trace.csv" -size +2M -delete
