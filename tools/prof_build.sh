#!/bin/bash
# Kernel trace of the GPU scene preparation on the 10 M-triangle height field (tools/bench_bigmesh.py --build device).
# usage: tools/prof_build.sh <outdir>
set -e
out=$1
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d "$out" -o build -- python3 tools/bench_bigmesh.py --build device --steps 1 > "$out/build.log" 2>&1
grep "rt_scene_create" "$out/build.log"
