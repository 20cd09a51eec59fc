#!/bin/bash
# Counters of one workload (tools/prof_config.py): memory-side traffic, L2 hits / misses, SQ; one rocprofv3 --pmc run
# per counter set.  usage: tools/pmc_config.sh <outdir> --config cX [prof_config args]
set -e
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${PMC_TIMEOUT:-200}
timeout -s KILL $T rocprofv3 --pmc FETCH_SIZE -d "$out" -o t1 -- python3 tools/prof_config.py --steps 1 "$@" > "$out/t1.log" 2>&1
timeout -s KILL $T rocprofv3 --pmc WRITE_SIZE -d "$out" -o t2 -- python3 tools/prof_config.py --steps 1 "$@" > "$out/t2.log" 2>&1
timeout -s KILL $T rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d "$out" -o t3 -- python3 tools/prof_config.py --steps 1 "$@" > "$out/t3.log" 2>&1
timeout -s KILL $T rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY -d "$out" -o s1 -- python3 tools/prof_config.py --steps 1 "$@" > "$out/s1.log" 2>&1
timeout -s KILL $T rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$out" -o s2 -- python3 tools/prof_config.py --steps 1 "$@" > "$out/s2.log" 2>&1
timeout -s KILL $T rocprofv3 --kernel-trace --stats -d "$out" -o kt -- python3 tools/prof_config.py --steps 3 "$@" > "$out/kt.log" 2>&1
grep -h "step" "$out"/t1.log "$out"/kt.log
# summaries (the result databases are tens of MB each: summarised, then dropped)
cfg=$(echo "$@" | sed -n 's/.*--config \([a-z0-9]*\).*/\1/p')
python3 tools/prof_summary.py "$out/kt_results.db" --kernel k_render --cmd "rocprofv3 --kernel-trace --stats -- python3 tools/prof_config.py --steps 3 $*" > "$out/kernel_trace.txt"
python3 tools/pmc_summary.py "$out" > "$out/pmc.txt"
python3 tools/traffic_json.py "$out" "$cfg" --out "$out/traffic_$cfg.json" > /dev/null
grep -h "step 0" "$out"/t1.log >> "$out/pmc.txt"
rm -f "$out"/*_results.db
