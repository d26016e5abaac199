#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out/r5_tile
timeout 1200 python -m pytest tests/test_lds_tile_hip.py -x -q -m gpu 2>&1 | tail -5
for lib in svae_amd/libsvae_hip.so tests/_variants/tile_unfused.so tests/_variants/tile_fused_dual.so; do
  for B in 512 256 64; do
    echo "== $lib B=$B"; SVAE_AMD_LIB=$lib timeout 300 python tools/bench_tile.py 64 1000 $B 4 2>&1 | tail -1
  done
done
SVAE_AMD_LIB=tests/_variants/tile_timing.so timeout 600 python tools/tile_timing_halves.py 64 1000 512 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_tile/timing_fused_b512.txt
