#!/bin/bash
# Memory-side traffic of one render launch: FETCH_SIZE and WRITE_SIZE (KB), one rocprofv3 --pmc run each (they
# do not fit one pass).  usage: tools/pmc_traffic.sh <outdir> [prof_render args]
set -e
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d "$out" -o t1 -- python3 tools/prof_render.py --steps 1 "$@" > "$out/t1.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d "$out" -o t2 -- python3 tools/prof_render.py --steps 1 "$@" > "$out/t2.log" 2>&1
grep "step 0" "$out"/t*.log
