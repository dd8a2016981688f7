#!/bin/bash
# dev (round 6): where the fused probe + threshold kernel's time goes.  Stop variants of the product kernel
# (rcorrector_amd/variants/fstop<N>.so = tools/build_k3_variant.sh fstop<N> -DRC_FUSED_STOP=<N>: the kernel returns after
# stage N, rc_quarter.h / rc_correct.hip) on the bench presets, `probe` = the kernel's ms per step (bench.py --no-extras),
# and SQ_INSTS_VALU of the same launches for a few of them.  Run on the GPU box from the repository root.
# Usage: tools/fused_stops.sh "<configs>" "<stops for every config>" "<stops for the pmc pass>"
cd "$(dirname "$0")/.."
CONFIGS=${1:-"2"}
STOPS=${2:-"0 1 2 3 10 11 12 13 14 15 16 17 18 19"}
PMC=${3:-""}
OUT=gpurun_out/r6_fused
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
line() {  # <config> <lib> <label>
  RC_LIB=$2 python bench.py --config $1 --no-extras --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']; k=c['kernel_ms_per_step']
print('config $1 %-10s probe %7.2f ms   single %6.2f correct %7.2f step %7.2f' % ('$3', k['probe'], k['single'], k['correct'], d['ms_per_step']))"
}
for c in $CONFIGS; do
  line $c rcorrector_amd/librcorrector_amd.so full
  for n in $STOPS; do
    [ -f rcorrector_amd/variants/fstop$n.so ] && line $c rcorrector_amd/variants/fstop$n.so stop$n
  done
  line $c rcorrector_amd/librcorrector_amd.so full
done | tee -a $OUT/stops_ms.txt
for n in $PMC; do
  lib=rcorrector_amd/variants/fstop$n.so
  [ "$n" = full ] && lib=rcorrector_amd/librcorrector_amd.so
  rm -rf /tmp/prof_pmc
  (cd /tmp && RC_LIB=$REPO/$lib timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --output-format csv -d /tmp/prof_pmc -- python $REPO/bench.py --config 2 --cpu-sample 0 --no-extras --steps 2 --warmup 0 > /dev/null 2> /tmp/prof_pmc.err)
  echo "== pmc stop $n"
  python tools/prof_summary.py pmc /tmp/prof_pmc | grep "^kernel\|k_probe"
done | tee -a $OUT/stops_pmc.txt
