#!/bin/bash
# Cache / memory counters of the BVH-bound mesh scene (tools/bench_mesh.py), one rocprofv3 --pmc run per pass.
# usage: tools/pmc_mesh.sh <outdir> [bench_mesh args]
set -e
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 tools/make_mesh_off.py /tmp/grid_225.off 225 > /dev/null
timeout -s KILL 180 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d "$out" -o m1 -- python3 tools/bench_mesh.py --steps 1 "$@" > "$out/m1.log" 2>&1
timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE -d "$out" -o m2 -- python3 tools/bench_mesh.py --steps 1 "$@" > "$out/m2.log" 2>&1
timeout -s KILL 180 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY -d "$out" -o m3 -- python3 tools/bench_mesh.py --steps 1 "$@" > "$out/m3.log" 2>&1
timeout -s KILL 180 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_REQ_sum GRBM_GUI_ACTIVE -d "$out" -o m4 -- python3 tools/bench_mesh.py --steps 1 "$@" > "$out/m4.log" 2>&1 || true
grep "step 0" "$out"/m*.log
