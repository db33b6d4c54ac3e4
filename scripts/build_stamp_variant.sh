#!/bin/bash
# Diagnostic build of libfte with in-kernel clock stamps (-DFTE_STAMP): variants/libfte_stamp.so.  NOT the product library.
set -e
HERE="$(cd "$(dirname "$0")/../tf_face_toolbox_amd/csrc" && pwd)"
OUT="$(cd "$(dirname "$0")/.." && pwd)/variants"
mkdir -p "$OUT" /tmp/fte_stamp_obj
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
$HIPCC $FLAGS -DFTE_STAMP -c "$HERE/igemm.hip" -o /tmp/fte_stamp_obj/igemm.o
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libfte_stamp.so" /tmp/fte_stamp_obj/igemm.o "$HERE/obj/igemm16.o" "$HERE/obj/wgrad16.o" "$HERE/obj/pw16.o" "$HERE/obj/kernels.o" "$HERE/obj/layers.o" "$HERE/obj/api.o"
echo "built $OUT/libfte_stamp.so"
