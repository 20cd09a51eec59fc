set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r1g
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r1g -o trace -- python3 bench.py --steps 5 --warmup 1 --cpu-spp 0 --flop-per-ray 488.97 > gpurun_out/r1g/bench_under_rocprof.json 2> gpurun_out/r1g/trace.err
bash tools/pmc_passes.sh gpurun_out/r1g/pmc > gpurun_out/r1g/pmc.log 2>&1
ls -R gpurun_out/r1g | head -30
