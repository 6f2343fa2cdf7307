#!/bin/bash
# One rank's share of a frame through the whole tiled path (render, gather, reassembly), rehearsed on one GPU for 2, 4 and 8
# ranks with two batches and with one batch in flight.  Usage (GPU box): bash tools/rank_share.sh <outdir> [config]
OUT=${1:-gpurun_out/rank_share}; CFG=${2:-3}; mkdir -p "$OUT"
for n in 2 4 8; do
  RT_BENCH_TILED=1 RT_BENCH_EMULATE_WORLD=$n timeout -k 10 300 python bench.py --config $CFG --steps 320 --warmup 32 --no-cpu-baseline > "$OUT/cfg${CFG}_share_$n.json" 2> "$OUT/log" || { tail -3 "$OUT/log"; exit 1; }
  RT_BENCH_TILED=1 RT_BENCH_EMULATE_WORLD=$n timeout -k 10 300 python bench.py --config $CFG --steps 320 --warmup 32 --no-cpu-baseline --frames-in-flight 1 > "$OUT/cfg${CFG}_share_${n}_one_in_flight.json" 2> "$OUT/log" || { tail -3 "$OUT/log"; exit 1; }
  python - "$OUT/cfg${CFG}_share_$n.json" "$OUT/cfg${CFG}_share_${n}_one_in_flight.json" <<'PY'
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "%.4f ms per frame, kernel %.4f ms per frame, batch %s, in flight %s" % (d["ms_per_step"], d["kernel_ms"], d["config"]["frames_per_gather"], d["config"]["frames_in_flight"]))
PY
done
