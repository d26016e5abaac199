#!/bin/bash
# N = 8 control-flow rehearsal on ONE GPU (gloo; NOT a scaling number) + N = 2, exactly as the driver launches bench.py
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO; mkdir -p gpurun_out/dp8
for N in 2 8; do
SVAE_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29611 \
   bench.py --gpus $N --steps 20 --warmup 3 > gpurun_out/dp8/bench_gloo_${N}ranks_1gpu.json 2> gpurun_out/dp8/bench_gloo_${N}ranks_1gpu.err
echo "N=$N rc=$?"; tail -c 400 gpurun_out/dp8/bench_gloo_${N}ranks_1gpu.json; echo
done
