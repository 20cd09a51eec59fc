#!/bin/bash
# Frame times with and without the treelet order of the wide nodes (RTAMD_TREELET_ORDER=0|1), the same library.
# usage (GPU box): bash tools/treelet_ab.sh <config> <steps>
cfg=$1; steps=$2
for round in 1 2; do for t in 0 1; do
  RTAMD_TREELET_ORDER=$t timeout -k 10 300 python3 tools/prof_config.py --config $cfg --tile 8 --steps $steps 2>&1 | sed -n 's/.*rays traced.*reference rays), \([0-9.]*\) ms.*/\1/p' | sort -n | head -3 | paste -sd' ' - | sed "s/^/treelet order $t (round $round, best three of $steps, ms): /"
done; done
