#!/bin/bash
# Full GPU test suite, then the per-generation times of the larger shapes: tools/gpu_check.sh <tag>
set -o pipefail
tag=${1:-chk}
out=gpurun_out
mkdir -p $out
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $out/${tag}_tests.log 2>&1
rc=$?
tail -5 $out/${tag}_tests.log
[ $rc -eq 0 ] || exit $rc
for c in 2 3 4 5; do
  timeout -k 10 200 python3 tools/concurrent_queries.py batch $c 1000 1 2>/dev/null | tee -a $out/${tag}_shapes.log || exit 1
done
