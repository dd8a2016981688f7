#!/bin/bash
# instruction-cache behaviour of the bench kernels
REPO=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM"; do
  rm -rf /tmp/prof_pmc
  timeout 400 rocprofv3 --pmc $C --output-format csv -d /tmp/prof_pmc -- python $REPO/bench.py --cpu-sample 0 --steps 2 --warmup 0 "$@" > /dev/null 2> /tmp/prof_pmc.err
  python $REPO/tools/prof_summary.py pmc /tmp/prof_pmc | grep -v "^kernel" | grep "k_correct\|k_thresh\|k_probe"
done
