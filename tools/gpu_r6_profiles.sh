#!/bin/bash
# Round-6 rocprofv3 evidence (every counter family in its own pass, never combined with a trace domain):
# headline kernel at 512 / 4096 sequences, SLDS ascent (producer-wavefront kernel + HMM), training path 512 / 4096,
# tile E-step n = 64, GMM step, end-to-end gradfun step.  Summaries: profiles/summarize.py / summarize_all.py.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
bash profiles/run_profile.sh r6_twoend 2>&1 | tail -1
( cd /tmp && export TMPDIR=/tmp; OUT=$REPO/gpurun_out/prof_r6_twoend
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -o bench -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra > $OUT/pmc_sq.log 2>&1 )
bash profiles/run_profile.sh r6_twoend_b4096 --seqs-per-gpu 4096 2>&1 | tail -1
bash profiles/run_profile.sh r6_tile_n64_b512 --workload lds64 --steps 3 --warmup 1 2>&1 | tail -1
( cd /tmp && export TMPDIR=/tmp; OUT=$REPO/gpurun_out/prof_r6_tile_n64_b512
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -o bench -- python $REPO/bench.py --workload lds64 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/pmc_sq.log 2>&1 )
bash tools/prof_generic.sh r6_slds python $REPO/tools/bench_slds.py 2048 500 10 8 --fused-only 2>&1 | tail -2
bash tools/prof_generic.sh r6_train python $REPO/tools/bench_train_path.py 512 200 10 1 2>&1 | tail -2
bash tools/prof_generic.sh r6_train_b4096 python $REPO/tools/bench_train_path.py 4096 200 10 1 2>&1 | tail -2
bash tools/prof_generic.sh r6_gmm python $REPO/tools/bench_gmm_step.py 2>&1 | tail -2
bash tools/prof_generic.sh r6_gradfun python $REPO/tools/bench_gradfun.py 2>&1 | tail -2
du -sh gpurun_out
# round 6: the wide HMM kernel (17 <= K <= 64) and the unprofiled run of the training-path A/B (lean vs full records)
( cd $REPO && timeout 300 python tools/bench_hmm.py --K 64 64 512 2048 > gpurun_out/prof_r6_hmm_wide.txt 2>&1; python tools/bench_hmm.py --K 24 2048 >> gpurun_out/prof_r6_hmm_wide.txt 2>&1; tail -3 gpurun_out/prof_r6_hmm_wide.txt )
( cd $REPO && timeout 300 python tools/bench_train_ab.py 4096 2048 1100 1024 512 > gpurun_out/prof_r6_train_ab.txt 2>&1; tail -15 gpurun_out/prof_r6_train_ab.txt )
