#!/bin/bash
# Diagnostic build of libfte with in-kernel clock stamps in wino_mm_kernel (-DFTE_WINO_STAMP): variants/libfte_wstamp.so.  NOT the product library.
set -e
HERE="$(cd "$(dirname "$0")/../../tf_face_toolbox_amd/csrc" && pwd)"
OUT="$(cd "$(dirname "$0")/../.." && pwd)/variants"
mkdir -p "$OUT" /tmp/fte_wstamp_obj
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
$HIPCC $FLAGS ${STAMP--DFTE_WINO_STAMP} $EXTRA -c "$HERE/wino.hip" -o /tmp/fte_wstamp_obj/wino.o
OBJS=""
for f in igemm igemm16 wgrad16 pw16 kernels layers api; do OBJS="$OBJS $HERE/obj/$f.o"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libfte_wstamp.so" /tmp/fte_wstamp_obj/wino.o $OBJS
echo "built $OUT/libfte_wstamp.so"
