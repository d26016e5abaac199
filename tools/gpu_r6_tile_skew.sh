#!/bin/bash
# round 6: skewed start of the two workgroups of a CU in the tile kernel (svae_amd/csrc/lds_estep_tile.hip, SVAE_TILE_SKEW_NS)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for ns in 0 4000 8000 11000 14000 18000; do
  echo "== skew $ns ns"
  SVAE_AMD_LIB=tests/_variants/tile_skew_$ns.so PYTHONPATH=. timeout 300 python tools/bench_tile.py 64 1000 512 6 2>&1 | tail -1
done
echo "== product library"
PYTHONPATH=. timeout 300 python tools/bench_tile.py 64 1000 512 6 2>&1 | tail -1
