set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r1h
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r1h -o trace -- python3 bench.py --steps 5 --warmup 1 --cpu-spp 0 --flop-per-ray 488.97 > gpurun_out/r1h/bench_under_rocprof.json 2> gpurun_out/r1h/trace.err
bash tools/pmc_passes.sh gpurun_out/r1h/pmc > gpurun_out/r1h/pmc.log 2>&1
ls -R gpurun_out/r1h | head -30
