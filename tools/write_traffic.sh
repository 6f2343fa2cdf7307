#!/bin/bash
# HBM write (and read) traffic per launch for library variants: one rocprofv3 pass each for WRITE_SIZE and FETCH_SIZE.
# Usage (GPU box, repo root): bash tools/write_traffic.sh "<variants>" [config]      ("tree" = the in-tree library)
VARIANTS=${1:-tree}; CFG=${2:-3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RT_SPLIT=0
for v in $VARIANTS; do
  if [ "$v" = tree ]; then unset RT_LIB_OVERRIDE; else export RT_LIB_OVERRIDE=$PWD/variants/$v.so; fi
  OUT=gpurun_out/wtraffic/$v-$CFG; rm -rf "$OUT"; mkdir -p "$OUT"
  for c in WRITE_SIZE FETCH_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d "$OUT/$c" -- python3 bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-second-leg --path-frames 0 > "$OUT/$c.log" 2>&1 || { echo "$v $c failed"; tail -3 "$OUT/$c.log"; }
  done
  echo "config $CFG $v: $(python3 tools/summarize_pmc.py "$OUT" | python3 -c 'import json,sys; d=json.load(sys.stdin); print("WRITE_SIZE %.1f MB, FETCH_SIZE x2 %.1f MB per launch" % (d.get("WRITE_SIZE",0)*1024/1e6, 2*d.get("FETCH_SIZE",0)*1024/1e6))')"
done
