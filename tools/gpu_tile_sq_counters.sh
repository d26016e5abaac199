#!/bin/bash
# Where the wavefront cycles of the tile E-step go (SQ counters, one pass):  bash tools/gpu_tile_sq_counters.sh [seqs-per-gpu]
# WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall: MFMA dependency, pipe busy) + ACTIVE_INST_ANY ~ WAVE_CYCLES
REPO=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-512}
OUT=$REPO/gpurun_out/sq_tile_b$B; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT -o sq -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --workload lds64 --seqs-per-gpu $B > $OUT/log.txt 2>&1
python - $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(sys.argv[1] + "/sq_counter_collection.csv")):
    k = row["Kernel_Name"]
    if "lds_estep_tile" not in k: continue
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    if row["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k, c in acc.items():
    w = c["SQ_WAVE_CYCLES"]
    print(k[:60], "launches", n[k])
    for name in sorted(c): print("   %-28s %.4g  (%.1f %% of wave cycles)" % (name, c[name] / max(1, n[k]), 100 * c[name] / w))
PY
