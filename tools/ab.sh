#!/bin/bash
# A/B runs of bench.py over library variants (variants/<name>.so, built by tools/build_variant.sh; "tree" = the in-tree library).
# Usage: bash tools/ab.sh "<variant names>" "<config numbers>" [steps]
VARIANTS=${1:-tree}; CONFIGS=${2:-3}; STEPS=${3:-200}
for cfg in $CONFIGS; do
  for round in 1 2; do
    for v in $VARIANTS; do
      if [ "$v" = tree ]; then unset RT_LIB_OVERRIDE; else export RT_LIB_OVERRIDE=$PWD/variants/$v.so; fi
      out=$(timeout -k 10 300 python bench.py --config $cfg --steps $STEPS --warmup 20 --no-cpu-baseline --no-second-leg --path-frames 0 2>/dev/null | tail -1)
      echo "config $cfg $v: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.4f ms/step, kernel %.4f ms, crc %s" % (d["ms_per_step"], d["kernel_ms"], d["check"]["frame_crc32"]))')"
    done
  done
done
