#!/bin/bash
# dev (round 6): where the fused probe + threshold kernel's time goes.  Stop variants of the product kernel
# (rcorrector_amd/variants/<prefix><N>.so = tools/build_k3_variant.sh <prefix><N> -DRC_FUSED_STOP=<N>: the kernel returns after
# stage N, rc_quarter.h / rc_correct.hip) on the bench presets; `probe` = the kernel's ms per step (bench.py --no-extras).
# Run on the GPU box from the repository root.
# Usage: tools/fused_stops.sh "<configs>" <prefix> "<stops>" [<library the stops are variants of>]
cd "$(dirname "$0")/.."
CONFIGS=${1:-"2"}
PREFIX=${2:-fstop}
STOPS=${3:-"1 2 3 10 12 13 14 16 17 18 19"}
FULL=${4:-rcorrector_amd/librcorrector_amd.so}
OUT=gpurun_out/r6_fused
mkdir -p $OUT
line() {  # <config> <lib> <label>
  RC_LIB=$2 python bench.py --config $1 --no-extras --steps 3 --warmup 1 --cpu-sample 0 2>$OUT/last.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']; k=c['kernel_ms_per_step']
print('config $1 %-10s probe %7.2f ms   single %6.2f correct %7.2f step %7.2f' % ('$3', k['probe'], k['single'], k['correct'], d['ms_per_step']))" || tail -3 $OUT/last.err
}
for c in $CONFIGS; do
  line $c $FULL $PREFIX-full
  for n in $STOPS; do
    [ -f rcorrector_amd/variants/$PREFIX$n.so ] && line $c rcorrector_amd/variants/$PREFIX$n.so $PREFIX$n
  done
  line $c $FULL $PREFIX-full
done | tee -a $OUT/stops_ms.txt
