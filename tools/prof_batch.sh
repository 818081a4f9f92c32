#!/bin/bash
# Per-kernel times of the replayed loop with Q queries in one context: tools/prof_batch.sh <cfg> <Q> <tag>
set -o pipefail
cfg=${1:-2}; Q=${2:-16}; tag=${3:-batch}
out=$PWD/gpurun_out/prof_${tag}
mkdir -p $out
export PYTHONPATH=$PWD; root=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 $root/tools/concurrent_queries.py batch $cfg 200 $Q > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
cd $root
grep "Q=" $out/run.log
python3 - $out/p_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r['Calls']) >= 50:
        print("%-100s %6s calls %8.2f us" % (r['Name'][:100], r['Calls'], float(r['AverageNs']) / 1000))
PY
