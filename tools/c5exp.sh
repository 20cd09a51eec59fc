cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2h
for env in "RTAMD_WIDE=0" "RTAMD_WIDE=1" "RTAMD_WIDE_NODES=12" "RTAMD_WIDE_NODES=6"; do
  echo "== $env"
  env $env timeout -k 10 200 python tools/bench_configs.py --config c5 --ranks 3 --pass-spp 64 2>&1 | grep "rank 3"
  env $env timeout -k 10 200 python tools/bench_configs.py --config c5 --ranks 3 2>&1 | grep "rank 3"
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r2h -o kt -- python3 tools/bench_configs.py --config c5 --ranks 3 > gpurun_out/r2h/kt.log 2>&1
python3 tools/prof_summary.py gpurun_out/r2h/kt_results.db --kernel k_ | head -40
rm -f gpurun_out/r2h/*.db
