#!/bin/bash
# Update-related GPU tests, then per-generation times of the benchmark shapes: tools/gpu_quick.sh <tag>
set -o pipefail
tag=${1:-q}
out=gpurun_out
mkdir -p $out
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py tests/test_gpu_queries.py tests/test_gpu_sharded_native.py -m gpu -x -q --timeout 300 --timeout-method thread > $out/${tag}_tests.log 2>&1
rc=$?
tail -5 $out/${tag}_tests.log
[ $rc -eq 0 ] || exit $rc
for c in 2 3 4 5; do
  timeout -k 10 200 python3 tools/concurrent_queries.py batch $c 2000 1 2>/dev/null | tee -a $out/${tag}_shapes.log || exit 1
done
timeout -k 10 120 python3 tools/diag_fused.py 2>/dev/null | tee $out/${tag}_diag_fused.log
