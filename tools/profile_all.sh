#!/bin/bash
# Kernel-trace statistics, the PMC passes and the derived traffic summary of the benchmark frame, all with block splitting off
# (under the profiler the selection would decide differently from the unprofiled bench).  Run on the GPU box from the repo root;
# copy the summaries to keep from gpurun_out/prof and gpurun_out/pmc into profiles/.
set -u
export RT_SPLIT=0
rm -rf gpurun_out/prof gpurun_out/pmc; mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/prof/bench.log 2>&1 || { tail -5 gpurun_out/prof/bench.log; exit 1; }
bash tools/profile_pmc.sh gpurun_out/pmc || exit 1
python3 tools/summarize_pmc.py gpurun_out/pmc > gpurun_out/pmc/summary.json
python3 tools/make_traffic.py gpurun_out/pmc/summary.json > gpurun_out/pmc/traffic.json
cat gpurun_out/pmc/traffic.json
find gpurun_out/prof -name "*kernel_stats.csv" | head -2
