#!/bin/bash
# Builds tools/math_check.hip with the kernel's own arithmetic flags (here, on the CPU) or runs it (on the GPU box).
# Usage: bash tools/math_check.sh build | run
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
case "$1" in
  build) mkdir -p "$ROOT/build" && /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fno-fast-math --offload-arch=gfx950 \
           -fhip-fp32-correctly-rounded-divide-sqrt -o "$ROOT/build/math_check" "$ROOT/tools/math_check.hip" ;;
  run) timeout -k 10 120 "$ROOT/build/math_check" ;;
  *) echo "usage: $0 build | run"; exit 2 ;;
esac
