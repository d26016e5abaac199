#!/bin/bash
# rocprofv3 evidence for ANY command, every counter family in its own pass (never combined with a trace domain):
#   tools/prof_generic.sh <tag> <command ...>      ->  gpurun_out/prof_<tag>/{trace,pmc_fetch,pmc_write,pmc_valu,pmc_sq}
# then (here or back in the container): python profiles/summarize_all.py <tag> [kernel-name substrings]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
(cd $REPO && python -c "from svae_amd import _lib; print(_lib.source_hash())") > $OUT/csrc_sha16.txt   # the sources profiled
cd /tmp && export TMPDIR=/tmp
timeout 400 "$@" > $OUT/bench.log 2>&1; tail -2 $OUT/bench.log | cut -c1-200
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- "$@" > $OUT/trace.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- "$@" > $OUT/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- "$@" > $OUT/pmc_write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d $OUT/pmc_valu -o bench -- "$@" > $OUT/pmc_valu.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -o bench -- "$@" > $OUT/pmc_sq.log 2>&1
find $OUT -name "*kernel_trace.csv" -size +1M -delete
find $OUT -name "*.csv" -size +30M -delete
ls $OUT/*/ | head -20
