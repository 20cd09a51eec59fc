#!/bin/bash
# SQ and cache counters of tools/bench_configs.py (one rocprofv3 --pmc run per counter set).
# usage: tools/pmc_configs.sh <outdir> [bench_configs args, e.g. --config c5 --ranks 3]
set -e
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 280 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY -d "$out" -o s1 -- python3 tools/bench_configs.py "$@" > "$out/s1.log" 2>&1
timeout -s KILL 280 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_INSTS_FLAT GRBM_GUI_ACTIVE -d "$out" -o s2 -- python3 tools/bench_configs.py "$@" > "$out/s2.log" 2>&1
timeout -s KILL 280 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d "$out" -o s3 -- python3 tools/bench_configs.py "$@" > "$out/s3.log" 2>&1
timeout -s KILL 280 rocprofv3 --pmc FETCH_SIZE -d "$out" -o s4 -- python3 tools/bench_configs.py "$@" > "$out/s4.log" 2>&1
grep "rank" "$out"/s*.log
