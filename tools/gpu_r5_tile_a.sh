#!/bin/bash
# round 5, tile path diagnosis: where the wavefronts of co-resident workgroups land + per-phase cycles of the halves
cd /root/repo
mkdir -p gpurun_out/r5_tile
timeout 120 tools/probes/hwid_probe > gpurun_out/r5_tile/hwid.txt 2>&1
echo "hwid rc=$?"; cat gpurun_out/r5_tile/hwid.txt
for B in 512 256; do
  SVAE_AMD_LIB=tests/_variants/tile_timing.so timeout 600 python tools/tile_timing_halves.py 64 1000 $B > gpurun_out/r5_tile/timing_b$B.txt 2>&1
  echo "timing B=$B rc=$?"; cat gpurun_out/r5_tile/timing_b$B.txt
done
