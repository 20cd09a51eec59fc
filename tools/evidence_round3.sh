#!/bin/bash
# The measurements of this round that are not rocprofv3 runs, each into a file of its own under gpurun_out/r3_misc/.
# usage (GPU box): bash tools/evidence_round3.sh
root=$(cd "$(dirname "$0")/.." && pwd); cd $root
out=gpurun_out/r3_misc; mkdir -p $out
set -e
timeout -k 10 300 python3 tools/step_overhead.py 8 > $out/c2_shares_of_eight.txt 2>&1
{ timeout -k 10 200 python3 tools/bench_trace.py --w 1920 --h 1080 --spp 4; timeout -k 10 200 python3 tools/bench_trace.py --w 960 --h 540 --spp 64; } > $out/bench_trace.txt 2>&1
{ for a in "" "--diffuse" "--leaf 1000" "--mesh 100"; do echo "# bench_kd.py $a"; timeout -k 10 200 python3 tools/bench_kd.py $a; done; } > $out/bench_kd.txt 2>&1
{ echo "# c5, one share of eight: four passes of 16 spp, then one pass of 64 spp"; timeout -k 10 300 python3 tools/bench_configs.py --config c5; timeout -k 10 300 python3 tools/bench_configs.py --config c5 --pass-spp 64; echo "# c4, one share of eight"; timeout -k 10 400 python3 tools/bench_configs.py --config c4 --ranks 0,3,7; } > $out/shares_c4_c5.txt 2>&1
bash tools/treelet_ab.sh c5 3 > $out/treelet_ab.txt 2>&1
bash tools/treelet_ab.sh c4 2 >> $out/treelet_ab.txt 2>&1
timeout -k 10 300 python3 bench.py > $out/bench_c2.json 2> $out/bench_c2.err
timeout -k 10 300 python3 bench.py --native-loop --cpu-spp 0 > $out/bench_c2_native_loop.json 2>> $out/bench_c2.err
for c in c3 c4 c5; do timeout -k 10 400 python3 bench.py --config $c --steps 3 --warmup 1 > $out/bench_$c.json 2>> $out/bench_c2.err; done
echo done
