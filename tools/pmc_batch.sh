#!/bin/bash
# Instruction-mix counters of the batched loop: tools/pmc_batch.sh <Q> <tag>
set -o pipefail
Q=${1:-64}; tag=${2:-pmcb}
out=$PWD/gpurun_out/pmc_${tag}
mkdir -p $out
export PYTHONPATH=$PWD; root=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $out -o p -- python3 $root/tools/concurrent_queries.py batch 2 60 $Q > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
cd $root
python3 tools/pmc_summary.py $out
