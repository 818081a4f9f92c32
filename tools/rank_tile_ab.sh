#!/bin/bash
# Tile length of the tiled ranking (tiles ranked by counting, then merged) at the larger shapes:
#   tools/rank_tile_ab.sh
set -o pipefail
for t in default 2048 1024; do
  if [ $t = default ]; then unset LMCMA_HIP_RANK_TILE; else export LMCMA_HIP_RANK_TILE=$t; fi
  for c in 3 4 5; do
    echo "tile $t cfg $c: $(timeout -k 10 100 python3 tools/concurrent_queries.py batch $c 1000 1 2>/dev/null)" || exit 1
  done
done
