# One profiling round of the Cornell frame: kernel trace of bench.py, SQ counter passes, memory traffic passes.
# usage (on the GPU box): bash tools/prof_round2.sh <tag>   -> gpurun_out/<tag>/
set -e
tag=${1:-r2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out -o trace -- python3 bench.py --steps 5 --warmup 1 --cpu-spp 0 --flop-per-ray 488.97 > $out/bench_under_rocprof.json 2> $out/trace.err
python3 tools/prof_summary.py $out/trace_results.db --kernel k_render --cmd "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 1 --cpu-spp 0 --flop-per-ray 488.97" > $out/kernel_trace.txt
bash tools/pmc_passes.sh $out/pmc > $out/pmc.log 2>&1
bash tools/pmc_traffic.sh $out/traffic > $out/traffic.log 2>&1
python3 tools/pmc_summary.py $out/pmc --rays 3098050155 --ms $(python3 -c "import json;print(json.load(open('$out/bench_under_rocprof.json'))['roofline']['kernel_ms'])") > $out/pmc_sq.txt
python3 tools/pmc_summary.py $out/traffic > $out/traffic.txt
python3 tools/traffic_json.py $out/traffic c2 --out $out/traffic_c2.json > /dev/null
find $out -name "*_results.db" -delete
cat $out/kernel_trace.txt | head -12; cat $out/pmc_sq.txt $out/traffic.txt
