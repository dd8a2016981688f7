#!/bin/bash
# dev: vector / scalar / LDS instruction counts of the probe kernel's launches, one library per argument (config 2, 2 untimed steps)
# Usage: tools/pmc_insts.sh <lib> [<lib> ...]      (run on the GPU box from the repository root)
cd "$(dirname "$0")/.."
REPO=$(pwd)
export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf /tmp/prof_pmc
  (cd /tmp && RC_LIB=$REPO/$lib timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --output-format csv -d /tmp/prof_pmc -- python $REPO/bench.py --config ${CONFIG:-2} --cpu-sample 0 --no-extras --steps 2 --warmup 0 > /dev/null 2> /tmp/prof_pmc.err)
  echo "== $lib"
  python tools/prof_summary.py pmc /tmp/prof_pmc | grep "^kernel\|k_probe\|k_single\|k_correct"
done
