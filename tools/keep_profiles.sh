#!/bin/bash
# Copy the summaries of one tools/prof_all.sh run from gpurun_out/ into profiles/ under the round's names.
# usage: bash tools/keep_profiles.sh <tag of the run> <prefix in profiles/>   e.g.  keep_profiles.sh r2f r2
set -e
tag=$1; pre=$2
src=gpurun_out/${tag}_c2
cp $src/kernel_trace.txt profiles/${pre}_c2_kernel_trace.txt
cp $src/bench_under_rocprof.json profiles/${pre}_c2_bench_under_rocprof.json
cp $src/pmc_sq.txt profiles/${pre}_c2_pmc_sq.txt
cp $src/traffic.txt profiles/${pre}_c2_traffic.txt
cp $src/traffic_c2.json profiles/traffic_c2.json
for c in c3 c4 c5; do
  src=gpurun_out/${tag}_$c
  cp $src/kernel_trace.txt profiles/${pre}_${c}_kernel_trace.txt
  cp $src/pmc.txt profiles/${pre}_${c}_pmc.txt
  cp $src/traffic_$c.json profiles/traffic_$c.json
done
