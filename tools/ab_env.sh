#!/bin/bash
# A/B runs of bench.py over settings of one environment variable (in-tree library).
# Usage: bash tools/ab_env.sh <VARIABLE> "<values>" "<config numbers>" [steps]      (value "-" = variable unset)
VAR=$1; VALUES=$2; CONFIGS=${3:-3}; STEPS=${4:-200}
for cfg in $CONFIGS; do
  for round in 1 2; do
    for v in $VALUES; do
      if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
      out=$(timeout -k 10 300 python bench.py --config $cfg --steps $STEPS --warmup 20 --no-cpu-baseline --no-second-leg --path-frames 0 2>/dev/null | tail -1)
      echo "config $cfg $VAR=$v: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.4f ms/step, kernel %.4f ms, crc %s" % (d["ms_per_step"], d["kernel_ms"], d["check"]["frame_crc32"]))')"
    done
  done
done
