#!/bin/bash
# Copies what tools/profile_round.sh <tag> left under gpurun_out/ into profiles/ (run here, after
# the GPU call):   tools/collect_round.sh <tag of the run> [<tag under profiles/>]
set -e
src=${1:?tag}
tag=${2:-$src}
out=gpurun_out
python3 profiles/collect.py $tag $out/${src}_trace $out/${src}_fetch $out/${src}_write
cp $out/${src}_bench.json profiles/${tag}_bench.json
cp $out/${src}_bench_20.json profiles/${tag}_bench_steps20.json
for c in 3 4 5; do
  cp $(find $out/${src}_c$c -name '*_kernel_stats.csv' | head -1) profiles/${tag}_cfg${c}_kernel_stats.csv
done
for q in 16 64; do
  cp $(find $out/${src}_q$q -name '*_kernel_stats.csv' | head -1) profiles/${tag}_cfg2_q${q}_kernel_stats.csv
done
cp $out/${src}_batched_queries.txt profiles/${tag}_batched_queries.txt
python3 profiles/summarize.py $out/${src}_trace $out/${src}_c3 $out/${src}_c4 $out/${src}_c5 $out/${src}_q16 $out/${src}_q64 > profiles/${tag}_summary.txt
cat profiles/${tag}_summary.txt
