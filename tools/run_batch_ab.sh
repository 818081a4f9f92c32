#!/bin/bash
# Batched queries with the one-workgroup ranking (from 16 queries) and with the counting rank only
set -o pipefail
for q in 16 32 64; do timeout -k 10 200 python3 tools/concurrent_queries.py batch 2 400 $q 2>/dev/null | tail -1 || exit 1; done
for q in 16 32 64; do LMCMA_HIP_SORTED_RANK_FROM=1000 timeout -k 10 200 python3 tools/concurrent_queries.py batch 2 400 $q 2>/dev/null | tail -1 || exit 1; done
