#!/bin/bash
# One rank's share of a frame through the whole tiled path (render, gather, reassembly), rehearsed on one GPU for 2, 4 and 8
# ranks with two batches and with one batch in flight, with tiles interleaved plainly (deal 1,1) and with rank 0's share
# reduced by what the reassembly costs it (deal auto).  Usage (GPU box): bash tools/rank_share.sh <outdir> [config]
OUT=${1:-gpurun_out/rank_share}; CFG=${2:-3}; mkdir -p "$OUT"
for n in 2 4 8; do
  for deal in 1,1 auto; do
    tag=cfg${CFG}_share_${n}_deal_${deal/,/_}
    RT_BENCH_TILED=1 RT_BENCH_EMULATE_WORLD=$n timeout -k 10 300 python bench.py --config $CFG --steps 320 --warmup 32 --no-cpu-baseline --path-frames 0 --tile-deal $deal > "$OUT/$tag.json" 2> "$OUT/log" || { tail -3 "$OUT/log"; exit 1; }
    RT_BENCH_TILED=1 RT_BENCH_EMULATE_WORLD=$n timeout -k 10 300 python bench.py --config $CFG --steps 320 --warmup 32 --no-cpu-baseline --path-frames 0 --tile-deal $deal --frames-in-flight 1 > "$OUT/${tag}_one_in_flight.json" 2> "$OUT/log" || { tail -3 "$OUT/log"; exit 1; }
    python - "$OUT/$tag.json" "$OUT/${tag}_one_in_flight.json" <<'PY'
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "%.4f ms per frame, kernel %.4f ms per frame, batch %s, in flight %s, deal %s, latency mode %s" % (
        d["ms_per_step"], d["kernel_ms"], d["config"]["frames_per_gather"], d["config"]["frames_in_flight"], d["config"].get("tile_deal"),
        d.get("ms_per_frame_latency")))
PY
  done
done
