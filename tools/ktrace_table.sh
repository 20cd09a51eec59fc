#!/bin/bash
# k_trace on the Cornell primary-ray batches under several builds of the library (tools/ab.sh build <name> -DRT_KTRACE_...):
# usage (GPU box): bash tools/ktrace_table.sh <name> ...      ("base" = the library proper)
root=$(cd "$(dirname "$0")/.." && pwd)
for round in 1 2; do for name in "$@"; do
  lib=$root/raytracing_amd/csrc/variants_$name.so
  [ "$name" = base ] && lib=$root/raytracing_amd/csrc/librtamd.so
  for shape in "--w 1920 --h 1080 --spp 4" "--w 960 --h 540 --spp 64"; do
    RTAMD_LIB=$lib timeout -k 10 200 python3 $root/tools/bench_trace.py $shape --only cornell | sed "s/^/$name ($shape): /"
  done
done; done
