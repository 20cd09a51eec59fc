# the mesh workloads in short: c3, c5 whole frames, c4 at 32 spp, 10 M-triangle full view; usage: meshexp.sh [ENV=VAL ...]
for e in "$@"; do export "$e"; done
for c in c3 c5; do timeout -k 10 300 python tools/prof_config.py --config $c --steps 2 2>&1 | grep "step 1" || exit 1; done
timeout -k 10 300 python tools/prof_config.py --config c4 --steps 2 --spp 32 2>&1 | grep "step 1" && timeout -k 10 100 python tools/bench_bigmesh.py --build device 2>&1 | grep "step 1"
