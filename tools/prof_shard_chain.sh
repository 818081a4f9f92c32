#!/bin/bash
# Per-kernel averages of one shard's generation with the direct exchange (images as peers): tools/prof_shard_chain.sh <G> <tag> [cfg] [nloc]
set -o pipefail
G=${1:-8}; tag=${2:-chain}; cfg=${3:-2}; nloc=${4:-}
out=$PWD/gpurun_out/prof_${tag}
mkdir -p $out
export PYTHONPATH=$PWD; cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 $OLDPWD/tools/shard_chain.py $G 1000 $cfg $nloc > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
cd $OLDPWD
tail -1 $out/run.log
f=$(ls $out/*kernel_stats.csv $out/*/*kernel_stats.csv 2>/dev/null | head -1)
cut -d, -f1-4 $f | cut -c1-150 | head -12
