#!/bin/bash
# Builds libfte.so (gfx950 only) next to the package.  Usage: build.sh [--clean] [outdir]
# The library carries the hash of the kernel sources it was built from (fte_version() ends in "src:<16 hex>", the same hash
# bench.py's kernel_src_sha() computes from csrc/*.hip, *.h): a stale binary cannot pass for the current sources.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
if [ "$1" = "--clean" ]; then rm -rf "$HERE/obj"; shift; fi
OUT="${1:-$HERE/..}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p "$HERE/obj"
SHA=$(cd "$HERE" && cat $(ls *.hip *.h | LC_ALL=C sort) | sha256sum | cut -c1-16)
if [ "$(cat "$HERE/obj/src_sha.txt" 2>/dev/null)" != "$SHA" ]; then rm -f "$HERE/obj/api.o"; fi      # api.o holds the stamp
pids=()
for f in igemm igemm16 wgrad16 pw16 kernels layers api; do
  if [ ! -f "$HERE/obj/$f.o" ] || [ "$HERE/$f.hip" -nt "$HERE/obj/$f.o" ] || [ "$HERE/igemm.h" -nt "$HERE/obj/$f.o" ] || [ "$HERE/igemm_dev.h" -nt "$HERE/obj/$f.o" ] \
     || [ "$HERE/kernels.h" -nt "$HERE/obj/$f.o" ] || [ "$HERE/layers.h" -nt "$HERE/obj/$f.o" ] || [ "$HERE/wgrad16.h" -nt "$HERE/obj/$f.o" ] || [ "$HERE/pw16.h" -nt "$HERE/obj/$f.o" ] \
     || [ "$HERE/../../include/fte.h" -nt "$HERE/obj/$f.o" ]; then
    $HIPCC $FLAGS -DFTE_SRC_SHA="\"$SHA\"" -c "$HERE/$f.hip" -o "$HERE/obj/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libfte.so" "$HERE/obj/igemm.o" "$HERE/obj/igemm16.o" "$HERE/obj/wgrad16.o" "$HERE/obj/pw16.o" "$HERE/obj/kernels.o" "$HERE/obj/layers.o" "$HERE/obj/api.o"
echo "$SHA" > "$HERE/obj/src_sha.txt"
echo "built $OUT/libfte.so (src:$SHA)"
