#!/bin/bash
# Stall-side SQ counters of the render kernel: scalar-memory latency, instruction fetch, branches, SALU cycles.
# usage: tools/pmc_stalls.sh <outdir> [prof_render args]
set -e
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_VALU_TRANS_F32 -d "$out" -o q1 -- python3 tools/prof_render.py --steps 1 "$@" > "$out/q1.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FLOPS_FP64 -d "$out" -o q2 -- python3 tools/prof_render.py --steps 1 "$@" > "$out/q2.log" 2>&1
grep "step 0" "$out"/q*.log
