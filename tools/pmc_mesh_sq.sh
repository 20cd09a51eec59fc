#!/bin/bash
# SQ counters of the mesh scene (tools/bench_mesh.py): instruction counts and waits of the BVH render variant.
# usage: tools/pmc_mesh_sq.sh <outdir> [bench_mesh args]
set -e
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 tools/make_mesh_off.py /tmp/grid_225.off 225 > /dev/null
timeout -s KILL 180 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY -d "$out" -o s1 -- python3 tools/bench_mesh.py --steps 1 "$@" > "$out/s1.log" 2>&1
timeout -s KILL 180 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_INSTS_FLAT GRBM_GUI_ACTIVE -d "$out" -o s2 -- python3 tools/bench_mesh.py --steps 1 "$@" > "$out/s2.log" 2>&1
grep "step 0" "$out"/s*.log
