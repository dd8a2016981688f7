#!/bin/bash
# bench lines of the length shapes (profiles/r4_len*_bench.json): 150 / 151 bases and the 90 / 9 / 1 % mix of 150 / 151 / 250, timed steps only
cd "$(dirname "$0")/.."
OUT=gpurun_out/r4m; mkdir -p $OUT
timeout 600 python bench.py --len 150 --no-extras --steps 5 --warmup 1 > $OUT/r4_len150_bench.json 2> /dev/null
timeout 600 python bench.py --len 151 --no-extras --steps 5 --warmup 1 > $OUT/r4_len151_bench.json 2> /dev/null
timeout 600 python bench.py --len-mix 150:0.9,151:0.09,250:0.01 --no-extras --steps 5 --warmup 1 > $OUT/r4_lenmix_90_9_1_bench.json 2> /dev/null
for f in len150 len151 lenmix_90_9_1; do python tools/fmt_bench.py $OUT/r4_${f}_bench.json | tail -1; done
