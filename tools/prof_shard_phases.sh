#!/bin/bash
# Per-kernel times of one rank's share of a G-GPU generation on a single GPU: tools/prof_shard_phases.sh <G> <tag>
set -o pipefail
G=${1:-8}; tag=${2:-shard}
out=$PWD/gpurun_out/prof_${tag}
mkdir -p $out
export PYTHONPATH=$PWD; cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 $OLDPWD/tools/shard_phases.py $G 400 > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
cd $OLDPWD
tail -1 $out/run.log
f=$(ls $out/*kernel_stats.csv $out/*/*kernel_stats.csv 2>/dev/null | head -1)
cut -d, -f1-4 $f | cut -c1-150 | head -12
LMCMA_HIP_LIB=$PWD/lmcma_path_planner_amd/lib/liblmcma_hip_diag.so timeout -k 10 100 python3 tools/shard_phases.py $G 50 2>&1 | tail -2
