#!/bin/bash
# Kernel traces at the larger shapes: tools/prof_shapes.sh <tag> [cfg ...]
set -o pipefail
tag=${1:-px}; shift
cfgs=${@:-3 4 5}
out=gpurun_out
export TMPDIR=/tmp
for c in $cfgs; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_c$c -o ${tag} -- python3 tools/concurrent_queries.py batch $c 300 1 > $out/${tag}_c$c.log 2>&1 || exit 1
done
python3 profiles/summarize.py $(for c in $cfgs; do echo $out/${tag}_c$c; done)
