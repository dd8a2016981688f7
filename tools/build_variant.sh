#!/bin/bash
# dev: build a differently configured library for A/B runs (tools/ab.sh, RC_LIB).
# Usage: tools/build_variant.sh <name> [extra hipcc flags, e.g. -DRC_K3_WAVES=4]
# Output: rcorrector_amd/variants/<name>.so (git-ignored; travels with gpurun)
set -eu
NAME=$1; shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
SRC=$REPO/rcorrector_amd/csrc
OUT=$REPO/rcorrector_amd/variants
mkdir -p "$OUT/obj_$NAME"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off $*"
for f in rc_api rc_api_table rc_api_batch rc_api_packed rc_table rc_transport rc_correct rc_correct_k23 rc_correct_k25 rc_correct_k31; do
  hipcc $FLAGS -c "$SRC/$f.hip" -o "$OUT/obj_$NAME/$f.o" &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/$NAME.so" "$OUT/obj_$NAME"/*.o
rm -rf "$OUT/obj_$NAME"
echo "$OUT/$NAME.so"
