#!/bin/bash
# Builds the library from the current state of cuda-ray-tracer_amd/csrc into variants/<name>.so (for A/B runs with
# RT_LIB_OVERRIDE); the in-tree objects are left alone.  Usage: bash tools/build_variant.sh <name> [extra hipcc flags]
# VARIANT_SED: an optional sed script applied to the copy of rt_kernels.hip first (throw-away experiments that do not belong
# in the source, e.g. a kernel that skips its frame stores to attribute HBM write traffic).
set -e
NAME=$1; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=/tmp/rt_variant_$NAME
rm -rf "$B"; mkdir -p "$B/cuda-ray-tracer_amd" "$ROOT/variants"
cp -r "$ROOT/cuda-ray-tracer_amd/csrc" "$B/cuda-ray-tracer_amd/csrc"
cp -r "$ROOT/include" "$B/include"
rm -f "$B"/cuda-ray-tracer_amd/csrc/*.o "$B"/cuda-ray-tracer_amd/csrc/*.so
if [ -n "${VARIANT_SED:-}" ]; then sed -i -E "$VARIANT_SED" "$B/cuda-ray-tracer_amd/csrc/rt_kernels.hip"; fi
make -C "$B/cuda-ray-tracer_amd/csrc" -j8 librt_amd.so CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. $*" > "$B/build.log" 2>&1 || { tail -20 "$B/build.log"; exit 1; }
cp "$B/cuda-ray-tracer_amd/csrc/librt_amd.so" "$ROOT/variants/$NAME.so"
echo "variants/$NAME.so"
