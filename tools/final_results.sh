#!/bin/bash
# Result files of all five benchmark configurations on the current build (default bench settings), then the profile of config 3.
# Run on the GPU box from the repo root: bash tools/final_results.sh <tag>; results land in gpurun_out/<tag>/.
TAG=${1:-final}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
for n in 1 2 3 4; do
  timeout -k 10 200 python bench.py --config $n > "$OUT/config$n.json" 2> "$OUT/config$n.log" || { echo "config $n failed"; tail -3 "$OUT/config$n.log"; exit 1; }
  echo "config $n done"
done
timeout -k 10 400 python bench.py --config 5 --detail 7 --path-frames 120 > "$OUT/config5.json" 2> "$OUT/config5.log" || { echo "config 5 failed"; tail -3 "$OUT/config5.log"; exit 1; }
echo "config 5 done"
bash tools/profile_all.sh $TAG/profile3 3 1920 1080
