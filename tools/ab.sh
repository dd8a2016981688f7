#!/bin/bash
# dev: A/B several builds of the library on the default bench workload (STEPS=n for more timed steps)
for lib in "$@"; do
  RC_LIB=$lib python bench.py $BENCH_ARGS --steps ${STEPS:-2} --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']; print('$lib reads/s=%.1fM step=%.1fms' % (d['value']/1e6, d['ms_per_step']), c['kernel_ms_per_step'])"
done
