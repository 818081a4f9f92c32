#!/bin/bash
# Kernel traces behind the LDS-or-L2 choice for the occupancy grid: tools/prof_grid_staging.sh <tag>
set -o pipefail
tag=${1:-gs}
out=gpurun_out
export TMPDIR=/tmp
for case in cfg2 vox64 vox64big; do
  for lds in 0 1; do
    export LMCMA_HIP_GRID_LDS=$lds
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_${case}_lds$lds -o ${tag} -- python3 tools/grid_staging.py run $case 300 > $out/${tag}_${case}_lds$lds.log 2>&1 || exit 1
    grep "us per generation" $out/${tag}_${case}_lds$lds.log
  done
done
unset LMCMA_HIP_GRID_LDS
timeout -k 10 200 python3 tools/grid_staging.py spans 2>/dev/null | tee $out/${tag}_spans.log
python3 profiles/summarize.py $out/${tag}_cfg2_lds0 $out/${tag}_cfg2_lds1 $out/${tag}_vox64_lds0 $out/${tag}_vox64_lds1 $out/${tag}_vox64big_lds0 $out/${tag}_vox64big_lds1
