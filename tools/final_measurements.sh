#!/bin/bash
# The measurement pass behind profiles/r2_*: kernel stats + counters for configs 1-2 (tools/profile_bench.sh),
# FETCH_SIZE / WRITE_SIZE passes for configs 3-4, profiles/r2_traffic.json from the FETCH_SIZE rows, then the four
# full bench lines and the end-to-end line.  Run on the GPU box from the repository root; results land in
# gpurun_out/ (copy what is to be kept into profiles/).
cd /root/repo
bash tools/profile_bench.sh r2_config1 1 --steps 3 --warmup 1
bash tools/profile_bench.sh r2_config2 2 --steps 3 --warmup 1
export TMPDIR=/tmp
for c in 3 4; do
  : > gpurun_out/r2_config${c}_bench_pmc.txt
  for C in "FETCH_SIZE" "WRITE_SIZE"; do
    rm -rf /tmp/prof_pmc
    (cd /tmp && timeout 900 rocprofv3 --pmc $C --kernel-include-regex "k_probe|k_threshold|k_correct" --output-format csv -d /tmp/prof_pmc -- python /root/repo/bench.py --config $c --no-extras --steps 1 --warmup 0 > /dev/null 2> /tmp/prof_pmc.err)
    python tools/prof_summary.py pmc /tmp/prof_pmc | grep -v "^kernel" | grep "k_probe\|k_thresh\|k_correct" >> gpurun_out/r2_config${c}_bench_pmc.txt
  done
done
# r2_traffic.json from the FETCH_SIZE rows of the probe kernel (per launch, KB)
python - <<'PY'
import json
out = {}
for c in (1, 2, 3, 4):
    kb = None
    for line in open("gpurun_out/r2_config%d_bench_pmc.txt" % c):
        f = line.split()
        if "FETCH_SIZE" in f and f[0].startswith("k_probe_threshold_list"):
            kb = float(f[f.index("FETCH_SIZE") + 2])   # mean over the launches
    out[str(c)] = {"k_probe_fetch_size_kb": kb, "k_probe_hbm_read_bytes_per_launch": kb * 1024 * 2,
                   "note": "2 x FETCH_SIZE (KB x 1024) of the probe kernel (k_probe_threshold_list), mean over the launches of one step, `rocprofv3 --pmc FETCH_SIZE --kernel-include-regex 'k_probe|k_threshold|k_correct' -- python bench.py --config %d --no-extras --steps 1 --warmup 0` (tools/profile_bench.sh; summary in profiles/r2_config%d_bench_pmc.txt); x2 because gfx950 tallies a 128-byte fabric request as 64 B" % (c, c)}
json.dump(out, open("gpurun_out/r2_traffic.json", "w"), indent=1)
print({k: v["k_probe_fetch_size_kb"] for k, v in out.items()})
PY
cp gpurun_out/r2_traffic.json profiles/r2_traffic.json
for c in 1 2 3 4; do
  timeout 1500 python bench.py --config $c > gpurun_out/r2_bench_config$c.json 2> gpurun_out/r2_bench_config$c.err
  python tools/fmt_bench.py gpurun_out/r2_bench_config$c.json
done
timeout 900 python bench.py --config 2 --e2e --no-extras 2>/dev/null | tail -1 > gpurun_out/r2_bench_e2e.json
tail -c 600 gpurun_out/r2_bench_e2e.json
