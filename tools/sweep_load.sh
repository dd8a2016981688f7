#!/bin/bash
# dev: K1/K3 time vs table load factor
for l in 0.30 0.45 0.55 0.65 0.75 0.85; do
  RC_TABLE_LOAD=$l python bench.py --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']; print('load=$l table=%.2fGiB reads/s=%.1fM probe=%.2fms thr=%.2fms cor=%.2fms frac=%.3f' % (c['table_bytes']/2**30, d['value']/1e6, c['kernel_ms_per_step']['probe'], c['kernel_ms_per_step']['threshold'], c['kernel_ms_per_step']['correct'], d['roofline']['frac']))"
done
