#!/bin/bash
# Vector / scalar instruction counts of the render kernel per frame (one rocprofv3 counter pass per configuration): the measure
# for changes that remove instructions, where frame times of different builds differ by code layout alone.
# Usage (GPU box, repo root): bash tools/count_insts.sh "<variants>" "<configs>"   (variant "tree" = the in-tree library)
VARIANTS=${1:-tree}; CONFIGS=${2:-3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RT_SPLIT=0
for cfg in $CONFIGS; do for v in $VARIANTS; do
  if [ "$v" = tree ]; then unset RT_LIB_OVERRIDE; else export RT_LIB_OVERRIDE=$PWD/variants/$v.so; fi
  OUT=gpurun_out/insts/$v-$cfg; rm -rf "$OUT"; mkdir -p "$OUT"
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d "$OUT/sq" -- python3 bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-second-leg --path-frames 0 > "$OUT/log" 2>&1 || { echo "$v $cfg failed"; tail -3 "$OUT/log"; continue; }
  echo "config $cfg $v: $(python3 tools/summarize_pmc.py "$OUT" | python3 -c 'import json,sys; d=json.load(sys.stdin); print("VALU %.3f M, SALU %.3f M per frame" % (d["SQ_INSTS_VALU"]/1e6, d["SQ_INSTS_SALU"]/1e6))')"
done; done
