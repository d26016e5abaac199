#!/bin/bash
# Round-4 rocprofv3 evidence, part B: every other kernel family with its counter passes (HBM bytes, VALU instructions,
# SQ wait / issue breakdown): training path (512 / 4096 sequences), SLDS ascent (fused LDS mean-field + HMM kernels),
# tile E-step (n = 64, 512 sequences), tile training pass (64 sequences), end-to-end gradfun step
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
bash tools/prof_generic.sh r4_train python $REPO/tools/bench_train_path.py 512 200 10 1 2>&1 | tail -2
bash tools/prof_generic.sh r4_train_b4096 python $REPO/tools/bench_train_path.py 4096 200 10 1 2>&1 | tail -2
bash tools/prof_generic.sh r4_slds python $REPO/tools/bench_slds.py 2048 500 10 8 --fused-only --run-inference 2>&1 | tail -2
bash tools/prof_generic.sh r4_tile_n64_b512 python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --workload lds64 2>&1 | tail -2
bash tools/prof_generic.sh r4_tile_train python $REPO/tools/bench_tile_train.py 64 1000 64 1 2>&1 | tail -2
