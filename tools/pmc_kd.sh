#!/bin/bash
# SQ counters of the kd-tree render variant (tools/bench_kd.py).  usage: tools/pmc_kd.sh <outdir> [bench_kd args]
set -e
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 180 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY -d "$out" -o k1 -- python3 tools/bench_kd.py --steps 1 "$@" > "$out/k1.log" 2>&1
timeout -s KILL 180 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT SQ_INSTS_BRANCH GRBM_GUI_ACTIVE -d "$out" -o k2 -- python3 tools/bench_kd.py --steps 1 "$@" > "$out/k2.log" 2>&1
grep "step 0" "$out"/k*.log
