#!/bin/bash
# The measurement pass behind profiles/r6_* (round 6; same as measure_r5.sh): for every preset the full bench line (live rocprofv3 --pmc passes inside bench.py,
# kept as gpurun_out/r6m/r6_traffic.json -- copy it to profiles/), the rocprofv3 kernel stats of `bench.py --no-extras` and the
# step timeline.  Run on the GPU box from the repository root.  Usage: tools/measure_r6.sh [configs, default "1 2 3 4"]
cd "$(dirname "$0")/.."
CONFIGS=${1:-"1 2 3 4"}
OUT=gpurun_out/r6m
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
[ -f profiles/r6_traffic.json ] && cp profiles/r6_traffic.json $OUT/r6_traffic.json
for c in $CONFIGS; do
  rm -rf /tmp/prof_stats
  (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $REPO/bench.py --config $c --no-extras --steps 3 --warmup 1 > $REPO/$OUT/r6_config${c}_bench_under_rocprof.json 2> /tmp/prof_stats.err)
  python tools/prof_summary.py stats /tmp/prof_stats > $OUT/r6_config${c}_bench_kernel_stats.txt
  python tools/step_timeline.py /tmp/prof_stats > $OUT/r6_config${c}_step_timeline.txt
  RC_BENCH_WRITE_TRAFFIC=$REPO/$OUT/r6_traffic.json RC_BENCH_PMC_TIMEOUT=1500 timeout 3000 python bench.py --config $c > $OUT/r6_bench_config$c.json 2> $OUT/r6_bench_config$c.err
  python tools/fmt_bench.py $OUT/r6_bench_config$c.json
done
