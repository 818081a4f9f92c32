#!/bin/bash
# Generations per recorded graph (LMCMA_HIP_GRAPH_GENS) against the bench line's rate, long and short runs:
#   tools/graph_len_ab.sh <tag>
set -o pipefail
tag=${1:-gl}
out=gpurun_out
mkdir -p $out
for g in default 8,2 32,8,2 100,20,4,1; do
  if [ $g = default ]; then unset LMCMA_HIP_GRAPH_GENS; else export LMCMA_HIP_GRAPH_GENS=$g; fi
  for a in "--steps 2000 --warmup 100" "--steps 20 --warmup 5" "--steps 20 --warmup 5"; do
    timeout -k 10 120 python3 bench.py $a --no-cpu-baseline --batched-queries 0 > $out/${tag}_tmp.json 2>> $out/${tag}_err.log || exit 1
    python3 -c "import json,sys; d=json.load(open('$out/${tag}_tmp.json')); print('$g', '$a', '%.4g /s  %.2f us/step wall  %.2f us device' % (d['value'], d['ms_per_step']*1e3, d['roofline']['avg_launch_us']))" | tee -a $out/${tag}_graph_len.txt
  done
done
