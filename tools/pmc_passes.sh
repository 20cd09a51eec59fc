#!/bin/bash
# Collect the SQ / GRBM counter passes for the render kernel (one rocprofv3 --pmc run per pass, as the
# hardware cannot hold more than 8 SQ counters at once).  usage: tools/pmc_passes.sh <outdir> [prof_render args]
set -e
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d "$out" -o p1 -- python3 tools/prof_render.py --steps 1 "$@" > "$out/p1.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE GRBM_COUNT -d "$out" -o p2 -- python3 tools/prof_render.py --steps 1 "$@" > "$out/p2.log" 2>&1
grep "step 0" "$out/p1.log" "$out/p2.log"
