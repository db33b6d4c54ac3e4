#!/bin/bash
# Builds libfte.so (gfx950 only) next to the package.  Usage: build.sh [outdir]
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="${1:-$HERE/..}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p "$HERE/obj"
pids=()
for f in igemm igemm16 wgrad16 pw16 kernels layers api; do
  if [ ! -f "$HERE/obj/$f.o" ] || [ "$HERE/$f.hip" -nt "$HERE/obj/$f.o" ] || [ "$HERE/igemm.h" -nt "$HERE/obj/$f.o" ] || [ "$HERE/igemm_dev.h" -nt "$HERE/obj/$f.o" ] \
     || [ "$HERE/kernels.h" -nt "$HERE/obj/$f.o" ] || [ "$HERE/layers.h" -nt "$HERE/obj/$f.o" ] || [ "$HERE/wgrad16.h" -nt "$HERE/obj/$f.o" ] || [ "$HERE/pw16.h" -nt "$HERE/obj/$f.o" ] \
     || [ "$HERE/../../include/fte.h" -nt "$HERE/obj/$f.o" ]; then
    $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$HERE/obj/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libfte.so" "$HERE/obj/igemm.o" "$HERE/obj/igemm16.o" "$HERE/obj/wgrad16.o" "$HERE/obj/pw16.o" "$HERE/obj/kernels.o" "$HERE/obj/layers.o" "$HERE/obj/api.o"
echo "built $OUT/libfte.so"
