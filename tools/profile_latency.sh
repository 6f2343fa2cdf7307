#!/bin/bash
# Where the render kernel waits: counter passes for the vector-memory path, LDS and instruction latencies (one rocprofv3 run
# per group, never combined with tracing).  Usage (GPU box, repo root): bash tools/profile_latency.sh <outdir> [bench args...]
set -u
OUT=${1:-gpurun_out/lat}; shift || true
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RT_SPLIT=0
run_pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-second-leg --path-frames 0 ${BENCH_ARGS:-} > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
  echo "pass $name ok"
}
run_pass sqA SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY
run_pass sqB SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
run_pass ta TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE
run_pass tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum
run_pass tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum
python3 tools/summarize_pmc.py "$OUT" > "$OUT/summary.json" && cat "$OUT/summary.json"
