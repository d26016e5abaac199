#!/bin/bash
# First contact with a multi-GPU MI355X node (no such node was available to rounds 1 - 5).
# On whatever node it lands: for N in {1,2,4,8} (capped at the GPUs present) run bench.py exactly as the driver does
# (one rank per GPU, torch.distributed.run, RCCL), once with the RCCL all-reduce and once with the opt-in IPC mailbox
# all-reduce (SVAE_BENCH_ALLREDUCE=mailbox); then tools/first_contact_check.py asserts, per N and per exchange,
#   * dist.get_world_size() == N with backend nccl,
#   * every rank's all-reduced statistics buffer is BIT-IDENTICAL (all_gather of the bytes) and equals the fp64 sum of the
#     per-rank buffers to rounding,
# and writes one JSON per (N, exchange) into gpurun_out/first_contact/.  Nothing here needs the reference or the network.
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO" || exit 1
OUT=gpurun_out/first_contact
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
echo "GPUs visible: $NGPU" | tee $OUT/summary.txt
PORT=29710
for N in 1 2 4 8; do
  [ "$N" -gt "$NGPU" ] && { echo "N=$N skipped (only $NGPU GPUs)" | tee -a $OUT/summary.txt; continue; }
  for EXCH in rccl mailbox; do
    [ "$N" -eq 1 ] && [ "$EXCH" = mailbox ] && continue
    PORT=$((PORT + 1))
    if [ "$EXCH" = mailbox ]; then export SVAE_BENCH_ALLREDUCE=mailbox; else unset SVAE_BENCH_ALLREDUCE; fi
    # (1) the exchange step checked: world size, backend, bit-identical reduced statistics on every rank
    timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
        tools/first_contact_check.py --out $OUT/check_${EXCH}_n${N}.json > $OUT/check_${EXCH}_n${N}.log 2>&1
    echo "N=$N $EXCH check rc=$?" | tee -a $OUT/summary.txt
    # (2) the bench line, as the driver launches it
    PORT=$((PORT + 1))
    if [ "$N" -eq 1 ]; then
      timeout 900 python bench.py --gpus 1 --steps 50 --warmup 5 --no-extra > $OUT/bench_${EXCH}_n${N}.json 2> $OUT/bench_${EXCH}_n${N}.err
    else
      timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
          bench.py --gpus $N --steps 50 --warmup 5 > $OUT/bench_${EXCH}_n${N}.json 2> $OUT/bench_${EXCH}_n${N}.err
    fi
    echo "N=$N $EXCH bench rc=$? $(tail -c 300 $OUT/bench_${EXCH}_n${N}.json | head -c 300)" | tee -a $OUT/summary.txt
  done
done
python - <<'PY' | tee -a gpurun_out/first_contact/summary.txt
import glob, json
rows = {}
for f in sorted(glob.glob("gpurun_out/first_contact/bench_*_n*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        rows[(f.split("_")[-2], d["n_gpus"])] = d["value"]
    except Exception as e:
        print(f, "unreadable:", e)
for (exch, n), v in sorted(rows.items()):
    base = rows.get((exch, 1)) or rows.get(("rccl", 1))
    print("%-8s N=%d  %.3e sequences/s  x%.2f of N=1" % (exch, n, v, v / base if base else float("nan")))
PY
