#!/bin/bash
# rocprofv3 passes over `bench.py --config <i> --no-extras` (timed steps only); summaries land in
# gpurun_out/ (copy the ones to keep into profiles/).  Usage: tools/profile_bench.sh <tag> <config> [bench.py args...]
# Counters are collected in their own runs (--pmc only, no trace domains), for the library's three
# per-batch kernels only: the generator's thousands of small torch kernels would each be serialised.
set -u
TAG=$1; CFG=$2; shift 2
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="--config $CFG --no-extras $*"
# 1. kernel trace + stats
rm -rf /tmp/prof_stats
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $REPO/bench.py $ARGS > $OUT/${TAG}_bench_under_rocprof.json 2> /tmp/prof_stats.err
python $REPO/tools/prof_summary.py stats /tmp/prof_stats > $OUT/${TAG}_bench_kernel_stats.txt
# 2. PMC passes
: > $OUT/${TAG}_bench_pmc.txt
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES"; do
  rm -rf /tmp/prof_pmc
  timeout 600 rocprofv3 --pmc $C --kernel-include-regex "k_probe|k_threshold|k_correct" --output-format csv -d /tmp/prof_pmc -- python $REPO/bench.py $ARGS --steps 1 --warmup 0 > /dev/null 2> /tmp/prof_pmc.err
  python $REPO/tools/prof_summary.py pmc /tmp/prof_pmc | grep -v "^kernel" | grep "k_probe\|k_thresh\|k_correct\|k_summary\|counter" >> $OUT/${TAG}_bench_pmc.txt
done
