#!/bin/bash
# Many partial rows summed to one row by column strips against the two-level sum (32 rows, then the
# update adds them), at the larger shapes: tools/reduce_ab.sh
set -o pipefail
for t in one-row two-levels; do
  if [ $t = one-row ]; then unset LMCMA_HIP_REDUCE_TWO_LEVELS; else export LMCMA_HIP_REDUCE_TWO_LEVELS=1; fi
  for c in 3 4 5; do
    echo "$t cfg $c: $(timeout -k 10 100 python3 tools/concurrent_queries.py batch $c 1000 1 2>/dev/null)" || exit 1
  done
done
