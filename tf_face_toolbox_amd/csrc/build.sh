#!/bin/bash
# Builds libfte.so (gfx950 only) next to the package.  Usage: build.sh [--clean] [outdir]
# The library carries the hash of the kernel sources it was built from (fte_version() ends in "src:<16 hex>", the same hash
# bench.py's kernel_src_sha() computes from csrc/*.hip, *.h and include/fte.h).  Every object is keyed by the hash of ITS source
# plus every header (obj/<name>.key): an object is reused only when that key matches, whatever the mtimes say -- so a stale object
# cannot link into a library stamped with the new hash.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
if [ "$1" = "--clean" ]; then rm -rf "$HERE/obj"; shift; fi
OUT="${1:-$HERE/..}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p "$HERE/obj"
FTEH="$HERE/../../include/fte.h"
SHA=$( (cd "$HERE" && cat $(ls *.hip *.h | LC_ALL=C sort)) | cat - "$FTEH" | sha256sum | cut -c1-16)
HDR=$( (cd "$HERE" && cat $(ls *.h | LC_ALL=C sort)) | cat - "$FTEH" | sha256sum | cut -c1-16)
SRCS="igemm igemm16 wgrad16 pw16 kernels layers wino api"
pids=()
for f in $SRCS; do
  KEY="$HDR-$(sha256sum < "$HERE/$f.hip" | cut -c1-16)-$FLAGS"
  if [ "$f" = api ]; then KEY="$KEY-$SHA"; fi          # api.o holds the stamp
  if [ ! -f "$HERE/obj/$f.o" ] || [ "$(cat "$HERE/obj/$f.key" 2>/dev/null)" != "$KEY" ]; then
    rm -f "$HERE/obj/$f.o" "$HERE/obj/$f.key"
    ( $HIPCC $FLAGS -DFTE_SRC_SHA="\"$SHA\"" -c "$HERE/$f.hip" -o "$HERE/obj/$f.o" && echo "$KEY" > "$HERE/obj/$f.key" ) &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
OBJS=""
for f in $SRCS; do OBJS="$OBJS $HERE/obj/$f.o"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libfte.so" $OBJS
echo "$SHA" > "$HERE/obj/src_sha.txt"
echo "built $OUT/libfte.so (src:$SHA)"
