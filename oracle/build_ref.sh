#!/usr/bin/env bash
# Builds oracle/_ref/ — the reference's OWN sources, compiled where they lie, as a checker for the CPU oracle (test
# infrastructure only; nothing in the product loads it).
#
#   oracle/_ref/libref.so      utils.cpp Sphere.cpp Plane.cpp Triangle.cpp Camera.cpp OBJ_Loader.cpp + oracle/ref_driver.cpp
#   oracle/_ref/ref_model_kat  Model.cpp utils.cpp OBJ_Loader.cpp + oracle/ref_model_driver.cpp (see that file's header)
#
# Nothing of the reference is copied into the repository and no header is written in its place:
#   <cuda_runtime.h>            the CUDA toolkit headers that ship inside the image's triton package
#   <gl/glew.h>, glm            the reference's own includes/ directory (-DGLEW_NO_GLU is glew's own switch for a system
#                               without GL/glu.h)
#   "glm\glm.hpp" (utils.h:9)   a Windows path; a symbolic link of that literal name, in the temporary build directory,
#                               pointing at the reference's own includes/glm/glm.hpp
# Objects are built in a temporary directory; only the two files above are written into the repository (oracle/_ref/ is
# git-ignored).  Not built, because they do not compile with g++: kernels.cu (needs <curand_kernel.h>, which the image does
# not have, and binds temporaries to `vec3&` at kernels.cu:67,130,152,188,224 — an MSVC-only extension) and AABB.cpp (the same
# extension at AABB.cpp:90).
set -euo pipefail

here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ref="${RT_REFERENCE_DIR:-/root/reference}"
cuda_inc="${RT_CUDA_INCLUDE:-}"
if [ -z "$cuda_inc" ]; then
    cuda_inc="$(python3 - <<'EOF'
import importlib.util, os
spec = importlib.util.find_spec("triton")
print(os.path.join(os.path.dirname(spec.origin), "backends", "nvidia", "include") if spec and spec.origin else "")
EOF
)"
fi
if [ ! -f "$ref/kernels.cu" ]; then echo "build_ref: no reference checkout at $ref" >&2; exit 3; fi
if [ ! -f "$cuda_inc/cuda_runtime.h" ]; then echo "build_ref: no cuda_runtime.h under '$cuda_inc'" >&2; exit 3; fi

out="$here/_ref"
mkdir -p "$out"
tmp="$(mktemp -d)"
trap 'rm -rf "$tmp"' EXIT
mkdir -p "$tmp/winpath"
ln -s "$ref/includes/glm/glm.hpp" "$tmp/winpath/glm\\glm.hpp"

CXX="${CXX:-g++}"
# -ffp-contract=off: plain IEEE binary32 in source order, the arithmetic contract the oracle and the kernels are held to
flags=(-std=c++17 -O2 -fPIC -ffp-contract=off -fno-fast-math -w -fpermissive -DGLEW_NO_GLU
       -I"$cuda_inc" -I"$ref/includes" -I"$ref/includes/glm" -I"$tmp/winpath" -I"$ref")

for unit in utils Sphere Plane Triangle Camera Model; do
    "$CXX" "${flags[@]}" -c "$ref/$unit.cpp" -o "$tmp/$unit.o" &
done
"$CXX" -std=c++17 -O2 -fPIC -ffp-contract=off -w -I"$ref" -c "$ref/OBJ_Loader.cpp" -o "$tmp/OBJ_Loader.o" &
"$CXX" "${flags[@]}" -c "$here/ref_driver.cpp" -o "$tmp/ref_driver.o" &
"$CXX" "${flags[@]}" -c "$here/ref_model_driver.cpp" -o "$tmp/ref_model_driver.o" &
fail=0
for job in $(jobs -p); do wait "$job" || fail=1; done
[ "$fail" = 0 ] || { echo "build_ref: compilation failed" >&2; exit 1; }

"$CXX" -shared -Wl,-z,defs -o "$out/libref.so.tmp" "$tmp/ref_driver.o" "$tmp/utils.o" "$tmp/Sphere.o" "$tmp/Plane.o" "$tmp/Triangle.o" \
    "$tmp/Camera.o" "$tmp/OBJ_Loader.o"
mv "$out/libref.so.tmp" "$out/libref.so"
# Model.o's references to AABB::*, Octree::Octree and cudaMallocManaged stay unresolved on purpose (never called)
"$CXX" -o "$out/ref_model_kat.tmp" "$tmp/ref_model_driver.o" "$tmp/Model.o" "$tmp/utils.o" "$tmp/OBJ_Loader.o" -Wl,--unresolved-symbols=ignore-all
mv "$out/ref_model_kat.tmp" "$out/ref_model_kat"
echo "build_ref: wrote $out/libref.so and $out/ref_model_kat"
