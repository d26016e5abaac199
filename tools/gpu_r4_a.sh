#!/bin/bash
# Round-4 first GPU call: the new parity tests (timed tile instances over long recursions, Schur-hint bit equality, VJP at
# 4096 sequences, SLDS at latent dim 16), then the default bench line as the round's starting point.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
timeout 1700 python -m pytest -m gpu -q -x -s \
  "tests/test_lds_tile_hip.py::test_tile_estep_timed_instances_full_length_against_reference" \
  "tests/test_lds_tile_hip.py::test_tile_estep_timed_instances_full_length_on_the_reference_generator" \
  "tests/test_lds_tile_hip.py::test_tile_estep_schur_scheduling_hint_does_not_change_a_bit" \
  "tests/test_vjp_hip.py::test_vjp_full_size_against_reference" \
  "tests/test_slds_hip.py::test_optimize_local_meanfield_at_latent_dim_16" \
  --durations=12 > gpurun_out/pytest_r4a.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error|vs compiled" gpurun_out/pytest_r4a.log | tail -12
grep -E "^FAILED|^ERROR" gpurun_out/pytest_r4a.log | head
python bench.py > gpurun_out/bench_r4a.json 2> gpurun_out/bench_r4a.err; tail -c 600 gpurun_out/bench_r4a.json
