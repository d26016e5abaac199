#!/bin/bash
# After tools/gpu_r6_profiles.sh (its raw output merged into gpurun_out/): the summaries committed under profiles/r6_*.
cd "$(dirname "$0")/.."
python profiles/summarize.py r6_twoend lds_estep_twoend_kernel | tail -2
python profiles/summarize_sq.py r6_twoend lds_estep_twoend_kernel | tail -1 | cut -c1-200
python profiles/summarize.py r6_twoend_b4096 lds_estep_twoend_rpc | tail -2
python profiles/summarize.py r6_tile_n64_b512 lds_estep_tile | tail -2
python profiles/summarize_sq.py r6_tile_n64_b512 "lds_estep_tile_kernel<4, false, 2, 1>" > /dev/null && mv profiles/r6_tile_n64_b512/sq_counters.json profiles/r6_tile_n64_b512/sq_counters_forward_half.json
python profiles/summarize_sq.py r6_tile_n64_b512 "lds_estep_tile_kernel<4, false, 2, 2>" > /dev/null && mv profiles/r6_tile_n64_b512/sq_counters.json profiles/r6_tile_n64_b512/sq_counters_backward_half.json
for t in r6_slds r6_train r6_train_b4096 r6_gmm r6_gradfun; do python profiles/summarize_all.py $t > gpurun_out/sum_$t.txt 2>&1; tail -1 gpurun_out/sum_$t.txt | cut -c1-160; done
python profiles/summarize_breakdown.py r6_gradfun | tail -1 | cut -c1-300
cp gpurun_out/prof_r6_train_ab.txt profiles/r6_train_b4096/ab_lean_vs_full.txt
cp gpurun_out/prof_r6_hmm_wide.txt profiles/r6_hmm_wide/bench_hmm.txt
grep -h csrc_sha16 profiles/r6_*/pmc*.json profiles/r6_*/sq_counters*.json | sort | uniq -c
