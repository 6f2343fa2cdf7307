#!/bin/bash
# Kernel-trace statistics, the PMC passes and the derived traffic summary of one benchmark configuration, all with block
# splitting off (under the profiler the selection would decide differently from the unprofiled bench) and without the
# bench's second leg and camera-path legs (every launch of the run is then of one kind).  Run on the GPU box from the repo root:
#   bash tools/profile_all.sh <tag> <config> <width> <height> [more bench.py arguments]
# Results land in gpurun_out/<tag>/; copy the summaries to keep into profiles/.
set -u
TAG=${1:-r02}; CONFIG=${2:-3}; W=${3:-1920}; H=${4:-1080}; shift 4 || true
export RT_SPLIT=0
export BENCH_ARGS="--config $CONFIG --no-second-leg --path-frames 0 $*"
OUT=gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT/prof"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 bench.py --steps ${PROFILE_STEPS:-300} --warmup 30 --no-cpu-baseline $BENCH_ARGS > "$OUT/bench.json" 2> "$OUT/bench.log" || { tail -5 "$OUT/bench.log"; exit 1; }
bash tools/profile_pmc.sh "$OUT/pmc" || exit 1
python3 tools/summarize_pmc.py "$OUT/pmc" > "$OUT/pmc_summary.json"
python3 tools/make_traffic.py "$OUT/pmc_summary.json" "config$CONFIG" "$W" "$H" > "$OUT/traffic.json"
cat "$OUT/traffic.json"
find "$OUT/prof" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
head -5 "$OUT/kernel_stats.csv"
