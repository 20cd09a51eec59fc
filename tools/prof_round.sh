set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r1f
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r1f -o trace -- python3 bench.py --steps 5 --warmup 1 --cpu-spp 0 --flop-per-ray 488.97 > gpurun_out/r1f/bench_under_rocprof.json 2> gpurun_out/r1f/trace.err
bash tools/pmc_passes.sh gpurun_out/r1f/pmc > gpurun_out/r1f/pmc.log 2>&1
ls -R gpurun_out/r1f | head -30
