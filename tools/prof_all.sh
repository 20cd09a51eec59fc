# Every committed profile of a round in one go (on the GPU box): the Cornell frame (prof_round2.sh) and the three mesh
# workloads (pmc_config.sh).  usage: bash tools/prof_all.sh <tag>  -> gpurun_out/<tag>_c2, <tag>_c3, <tag>_c4, <tag>_c5
tag=${1:-r2}
bash tools/prof_round2.sh ${tag}_c2 > gpurun_out/${tag}_c2.log 2>&1 || exit 1
echo "c2 done"
for c in c3 c5 c4; do
  PMC_TIMEOUT=280 bash tools/pmc_config.sh gpurun_out/${tag}_$c --config $c > gpurun_out/${tag}_$c.log 2>&1 || exit 1
  echo "$c done"
done
du -sh gpurun_out
