#!/bin/bash
# dev: instruction / wait counters of k_correct for one build of the library
# Usage: tools/pmc_k3.sh <lib.so> [bench args]   (counters only, k_correct only, one step)
LIB=$1; shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp RC_LIB=$LIB
cd /tmp
echo "== $LIB"
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES"; do
  rm -rf /tmp/prof_pmc
  timeout 400 rocprofv3 --pmc $C --kernel-include-regex "k_correct" --output-format csv -d /tmp/prof_pmc -- python $REPO/bench.py --config 2 --no-extras --steps 1 --warmup 0 "$@" > /dev/null 2> /tmp/prof_pmc.err
  python $REPO/tools/prof_summary.py pmc /tmp/prof_pmc | grep "k_correct" | awk '{printf "%-28s %18.0f\n", $2, $NF}'
done
