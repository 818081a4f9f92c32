#!/bin/bash
# Workgroup size of the counting rank at the larger shapes (64 candidates per workgroup halve the
# partial rows the update reads): tools/rank_threads_ab.sh
set -o pipefail
for t in default 1024; do
  if [ $t = default ]; then unset LMCMA_HIP_RANK_THREADS; else export LMCMA_HIP_RANK_THREADS=$t; fi
  for c in 3 4 5; do
    echo "rank threads $t cfg $c: $(timeout -k 10 100 python3 tools/concurrent_queries.py batch $c 1000 1 2>/dev/null)" || exit 1
  done
done
