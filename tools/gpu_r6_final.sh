#!/bin/bash
# Round-6 closing run: the whole GPU suite, smoke, the multi-rank rehearsals on one GPU, first-contact at N = 1, the default
# bench line (with traffic from the committed r5 profiles when their source hash matches this build)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO; mkdir -p gpurun_out/r6_final
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -v "^\s*$" > gpurun_out/r6_final/pytest_gpu.log; echo "pytest rc=${PIPESTATUS[0]}"
tail -3 gpurun_out/r6_final/pytest_gpu.log; grep -E "^FAILED|^ERROR" gpurun_out/r6_final/pytest_gpu.log | head -20
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
for N in 2 8; do
  SVAE_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2961$N \
     bench.py --gpus $N --steps 20 --warmup 3 > gpurun_out/r6_final/bench_gloo_${N}ranks_1gpu.json 2> gpurun_out/r6_final/bench_gloo_${N}ranks_1gpu.err
  echo "rehearsal N=$N rc=$? $(tail -c 300 gpurun_out/r6_final/bench_gloo_${N}ranks_1gpu.json | head -c 200)"
done
# round 6: `python bench.py --gpus 2` WITHOUT a launcher must bring up its own two ranks (gloo: one GPU on this box)
SVAE_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 3 --no-extra --no-cpu-baseline > gpurun_out/r6_final/bench_selflaunch_2ranks_1gpu.json 2> gpurun_out/r6_final/bench_selflaunch_2ranks_1gpu.err
echo "self-launch N=2 rc=$? $(tail -c 400 gpurun_out/r6_final/bench_selflaunch_2ranks_1gpu.json | head -c 300)"
timeout 900 bash tools/first_contact_8gpu.sh > gpurun_out/r6_final/first_contact.log 2>&1; echo "first contact rc=$?"; tail -6 gpurun_out/r6_final/first_contact.log
timeout 900 python bench.py > gpurun_out/r6_final/bench.json 2> gpurun_out/r6_final/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6_final/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "parity", d["parity"]["max_rel"], d["parity"]["ok"])
print("roofline", {k: d["roofline"][k] for k in ("frac", "kernel_ms", "traffic", "traffic_source", "traffic_over_algorithmic")})
for i, e in enumerate(d.get("extra", [])):
    print(i, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in e.items() if k in ("value", "ms_per_step", "ms_per_pass", "ms_per_ascent", "ms_per_run_inference", "us_per_fixed_point", "us_per_step", "us_per_step_graph", "eager_ms_per_step", "graph_ms_per_step", "error")},
          (e.get("roofline") or {}).get("traffic"), (e.get("parity") or {}).get("max_rel"))
print("cpu", {k: d["cpu_baseline"].get(k) for k in ("value", "cores", "kind")})
PY
