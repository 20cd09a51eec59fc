#!/bin/bash
# One profiling round of one workload: kernel trace of bench.py, then one rocprofv3 --pmc run per counter set over one frame of
# the same workload (tools/prof_config.py, the tile size bench.py uses), summarised into counters_<config>.json.
# usage (on the GPU box): bash tools/prof_round3.sh <tag> <c1..c5> [bench steps]   -> gpurun_out/<tag>_<config>/
set -e
tag=${1:-r3}; cfg=${2:-c2}; steps=${3:-5}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${tag}_$cfg
mkdir -p $out
T=${PMC_TIMEOUT:-200}
cmd="python3 bench.py --config $cfg --steps $steps --warmup 1 --cpu-spp 0"
timeout -k 10 $T rocprofv3 --kernel-trace --stats -d $out -o trace -- $cmd > $out/bench_under_rocprof.json 2> $out/trace.err
python3 tools/prof_summary.py $out/trace_results.db --kernel k_render --cmd "rocprofv3 --kernel-trace --stats -- $cmd" > $out/kernel_trace.txt
run="python3 tools/prof_config.py --config $cfg --tile 8 --steps 1"
timeout -s KILL $T rocprofv3 --pmc FETCH_SIZE -d $out -o t1 -- $run > $out/t1.log 2>&1
timeout -s KILL $T rocprofv3 --pmc WRITE_SIZE -d $out -o t2 -- $run > $out/t2.log 2>&1
timeout -s KILL $T rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $out -o t3 -- $run > $out/t3.log 2>&1
timeout -s KILL $T rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY -d $out -o s1 -- $run > $out/s1.log 2>&1
timeout -s KILL $T rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE -d $out -o s2 -- $run > $out/s2.log 2>&1
rays=$(sed -n 's/.*: \([0-9.]*\) rays traced.*/\1/p' $out/t1.log | head -1)
ms=$(sed -n 's/.*reference rays), \([0-9.]*\) ms.*/\1/p' $out/t1.log | head -1)
python3 tools/pmc_summary.py $out > $out/pmc.txt
grep -h "step 0" $out/t1.log >> $out/pmc.txt
python3 tools/counters_json.py $out $cfg --traced-rays ${rays:-0} --frame-ms ${ms:-0} --out $out/counters_$cfg.json > /dev/null
rm -f $out/*_results.db
head -8 $out/kernel_trace.txt; cat $out/bench_under_rocprof.json; cat $out/counters_$cfg.json
