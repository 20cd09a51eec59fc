#!/bin/bash
# A/B builds of the library and their timing.
#   tools/ab.sh build <name> [hipcc flags ...]   -> raytracing_amd/csrc/variants_<name>.so (here, no GPU; RT_EXP_* switches allowed)
#   tools/ab.sh run <config> <steps> <name> ...   (on the GPU box) every variant twice in turn through tools/prof_config.py;
#                                                 "base" = the library proper.  Prints the frame times of every run.
# Where a code object is loaded moves a frame time by up to 1.5 %: only differences beyond that count.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mode=$1; shift
if [ "$mode" = build ]; then
  name=$1; shift
  make -s -C $root/raytracing_amd/csrc OUT=variants_$name.so EXTRA="-DRT_ATTRIBUTION_BUILD $*"
  exit 0
fi
cfg=$1; steps=$2; shift 2
for round in 1 2; do
  for name in "$@"; do
    lib=$root/raytracing_amd/csrc/variants_$name.so
    [ "$name" = base ] && lib=$root/raytracing_amd/csrc/librtamd.so
    RTAMD_LIB=$lib timeout -k 10 300 python3 $root/tools/prof_config.py --config $cfg --tile 8 --steps $steps 2>&1 | sed -n 's/.*rays traced.*reference rays), \([0-9.]*\) ms.*/\1/p' | sort -n | head -3 | paste -sd' ' - | sed "s/^/$name (round $round, best three of $steps, ms): /"
  done
done
