#!/bin/bash
# Collects one round's measurements on the GPU box into gpurun_out/<tag>_*:
#   tools/profile_round.sh <tag>
# bench line, kernel trace of the same command, the two counter passes behind the HBM
# traffic figure (each in a run of its own), kernel traces at the larger shapes.
# Every GPU step has its own time limit and the chain stops at the first failure.
set -o pipefail
tag=${1:-rXX}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
py=python3
timeout -k 10 400 $py bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err &&
timeout -k 10 200 $py bench.py --steps 20 --warmup 5 --no-cpu-baseline --batched-queries 0 > $out/${tag}_bench_20.json 2>> $out/${tag}_bench.err &&
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -o ${tag} -- $py bench.py --no-cpu-baseline --batched-queries 0 > $out/${tag}_trace.log 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch -o ${tag} -- $py bench.py --steps 200 --warmup 20 --no-cpu-baseline --batched-queries 0 > $out/${tag}_fetch.log 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_write -o ${tag} -- $py bench.py --steps 200 --warmup 20 --no-cpu-baseline --batched-queries 0 > $out/${tag}_write.log 2>&1 &&
for c in 3 4 5; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_c$c -o ${tag} -- $py tools/concurrent_queries.py batch $c 300 1 > $out/${tag}_c$c.log 2>&1 || exit 1
done &&
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_q16 -o ${tag} -- $py tools/concurrent_queries.py batch 2 300 16 > $out/${tag}_q16.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_q64 -o ${tag} -- $py tools/concurrent_queries.py batch 2 300 64 > $out/${tag}_q64.log 2>&1 &&
timeout -k 10 300 $py tools/concurrent_queries.py batch 2 400 1 2 4 8 16 32 64 > $out/${tag}_batched_queries.txt 2> /dev/null &&
$py profiles/summarize.py $out/${tag}_trace $out/${tag}_c3 $out/${tag}_c4 $out/${tag}_c5 $out/${tag}_q16 $out/${tag}_q64
