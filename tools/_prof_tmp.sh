set -u
export RT_SPLIT=0
rm -rf gpurun_out/prof_e gpurun_out/pmc_e; mkdir -p gpurun_out/prof_e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_e -- python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/prof_e/bench.log 2>&1 || { tail -5 gpurun_out/prof_e/bench.log; exit 1; }
bash tools/profile_pmc.sh gpurun_out/pmc_e || exit 1
python3 tools/summarize_pmc.py gpurun_out/pmc_e > gpurun_out/pmc_e/summary.json
python3 tools/make_traffic.py gpurun_out/pmc_e/summary.json > gpurun_out/pmc_e/traffic.json
cat gpurun_out/pmc_e/traffic.json
find gpurun_out/prof_e -name "*kernel_stats.csv" | head -2
