#!/bin/bash
# Memory-side traffic and L2 behaviour of the render kernel on the 10 M-triangle scene (tools/bench_bigmesh.py):
# FETCH_SIZE, WRITE_SIZE and TCC hit / miss counts, one rocprofv3 --pmc run each.
# usage: tools/pmc_bigmesh.sh <outdir> [bench_bigmesh args]
set -e
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d "$out" -o f1 -- python3 tools/bench_bigmesh.py --steps 1 "$@" > "$out/f1.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d "$out" -o f2 -- python3 tools/bench_bigmesh.py --steps 1 "$@" > "$out/f2.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES -d "$out" -o f3 -- python3 tools/bench_bigmesh.py --steps 1 "$@" > "$out/f3.log" 2>&1
grep "step 0" "$out"/f*.log
