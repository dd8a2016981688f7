#!/bin/bash
# dev: a variant of the library in which ONE translation unit is recompiled with extra flags (the other objects come from
# the in-tree build).  Usage: tools/build_variant2.sh <name> <unit, e.g. rc_correct_k23> [extra hipcc flags]
# -> rcorrector_amd/variants/<name>.so; run a bench with it through RC_LIB=<path> (tools/ab.sh).
set -eu
NAME=$1; UNIT=$2; shift 2
REPO=$(cd "$(dirname "$0")/.." && pwd)
SRC=$REPO/rcorrector_amd/csrc
OUT=$REPO/rcorrector_amd/variants
mkdir -p "$OUT"
OBJS="rc_api rc_api_table rc_api_batch rc_api_packed rc_table rc_transport rc_correct rc_correct_k23 rc_correct_k25 rc_correct_k31"
make -s -C "$SRC" $(for o in $OBJS; do echo $o.o; done) >/dev/null
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off "$@" -c "$SRC/$UNIT.hip" -o "$OUT/$NAME.$UNIT.o"
LINK=""
for o in $OBJS; do if [ "$o" = "$UNIT" ]; then LINK="$LINK $OUT/$NAME.$UNIT.o"; else LINK="$LINK $SRC/$o.o"; fi; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/$NAME.so" $LINK
rm -f "$OUT/$NAME.$UNIT.o"
echo "$OUT/$NAME.so"
