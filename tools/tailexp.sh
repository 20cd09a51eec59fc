# tail-related timings: c5 share (one pass / four passes), kd scene, Cornell frame and shares; usage: tailexp.sh <outdir> [ENV=VAL ...]
out=$1; shift
mkdir -p $out
for e in "$@"; do export "$e"; done
timeout -k 10 200 python tools/bench_configs.py --config c5 --ranks 0,3,7 > $out/c5_4x16.txt 2>&1 && timeout -k 10 200 python tools/bench_configs.py --config c5 --ranks 0,3,7 --pass-spp 64 > $out/c5_onepass.txt 2>&1 && timeout -k 10 100 python tools/bench_kd.py > $out/kd.txt 2>&1 && timeout -k 10 200 python tools/step_overhead.py > $out/step.txt 2>&1
grep -h "rank" $out/c5_4x16.txt $out/c5_onepass.txt; tail -1 $out/kd.txt; grep world $out/step.txt
