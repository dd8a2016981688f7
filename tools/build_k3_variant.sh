#!/bin/bash
# dev: like build_variant.sh, but only rc_correct.hip is recompiled (the other objects come from the
# in-tree build).  Usage: tools/build_k3_variant.sh <name> [extra hipcc flags, e.g. -DRC_EXP_STOP=3]
set -eu
NAME=$1; shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
SRC=$REPO/rcorrector_amd/csrc
OUT=$REPO/rcorrector_amd/variants
mkdir -p "$OUT"
make -s -C "$SRC" rc_api.o rc_api_table.o rc_api_batch.o rc_api_packed.o rc_table.o rc_transport.o rc_correct_k23.o rc_correct_k25.o rc_correct_k31.o >/dev/null
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off "$@" -c "$SRC/rc_correct.hip" -o "$OUT/$NAME.rc_correct.o"
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/$NAME.so" "$SRC/rc_api.o" "$SRC/rc_api_table.o" "$SRC/rc_api_batch.o" "$SRC/rc_api_packed.o" "$SRC/rc_table.o" "$SRC/rc_transport.o" "$SRC/rc_correct_k23.o" "$SRC/rc_correct_k25.o" "$SRC/rc_correct_k31.o" "$OUT/$NAME.rc_correct.o"
rm -f "$OUT/$NAME.rc_correct.o"
echo "$OUT/$NAME.so"
