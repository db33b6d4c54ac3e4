#!/bin/bash
# A/B builds of libfte with extra compile flags for igemm.hip / igemm16.hip: variants/libfte_<name>.so (same ABI; select with FTE_LIB).
#   scripts/build_variant.sh noprio "-DFTE_PRIO_PROLOGUE=0 -DFTE_PRIO_EPILOGUE=0"
set -e
NAME="$1"; EXTRA="$2"
HERE="$(cd "$(dirname "$0")/../tf_face_toolbox_amd/csrc" && pwd)"
OUT="$(cd "$(dirname "$0")/.." && pwd)/variants"
OBJ=/tmp/fte_variant_$NAME
mkdir -p "$OUT" "$OBJ"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
$HIPCC $FLAGS $EXTRA -c "$HERE/igemm.hip" -o "$OBJ/igemm.o" &
$HIPCC $FLAGS $EXTRA -c "$HERE/igemm16.hip" -o "$OBJ/igemm16.o" &
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libfte_$NAME.so" "$OBJ/igemm.o" "$OBJ/igemm16.o" "$HERE/obj/wgrad16.o" "$HERE/obj/kernels.o" "$HERE/obj/layers.o" "$HERE/obj/api.o"
echo "built $OUT/libfte_$NAME.so"
