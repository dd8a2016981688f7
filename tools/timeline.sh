#!/bin/bash
# kernel timeline of the last step of `bench.py --no-extras <args>` (tools/step_timeline.py) + the kernel stats; on the GPU box
# usage: tools/timeline.sh <tag> [bench.py args...]   -> gpurun_out/<tag>_timeline.txt, gpurun_out/<tag>_kernel_stats.txt
TAG=$1; shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_tl
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tl -- python $REPO/bench.py --no-extras --steps 3 --warmup 1 "$@" > $OUT/${TAG}_bench_under_rocprof.json 2> /tmp/prof_tl.err
python $REPO/tools/step_timeline.py /tmp/prof_tl > $OUT/${TAG}_timeline.txt
python $REPO/tools/prof_summary.py stats /tmp/prof_tl > $OUT/${TAG}_kernel_stats.txt
