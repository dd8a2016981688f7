#!/bin/bash
# The measurement pass behind profiles/r3_*: rocprofv3 kernel stats for configs 1-2, counter passes (FETCH_SIZE; TCC; SQ
# instruction counts -- counters only, one group per run, restricted to the library's per-batch kernels) for configs 1-4,
# profiles/r3_traffic.json from them (with the git blob hashes of the kernel sources: bench.py reports a figure only while
# they match), then the four full bench lines.  Run on the GPU box from the repository root; results land in gpurun_out/r3m/
# (copy what is to be kept into profiles/).  Usage: tools/measure_r3.sh [configs, default "1 2 3 4"]
cd "$(dirname "$0")/.."
CONFIGS=${1:-"1 2 3 4"}
OUT=gpurun_out/r3m
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
for c in $CONFIGS; do
  if [ $c -le 2 ]; then
    rm -rf /tmp/prof_stats
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $REPO/bench.py --config $c --no-extras --steps 3 --warmup 1 > $REPO/$OUT/r3_config${c}_bench_under_rocprof.json 2> /tmp/prof_stats.err)
    python tools/prof_summary.py stats /tmp/prof_stats > $OUT/r3_config${c}_bench_kernel_stats.txt
  fi
  : > $OUT/r3_config${c}_bench_pmc.txt
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES"; do
    rm -rf /tmp/prof_pmc
    (cd /tmp && timeout 1200 rocprofv3 --pmc $C --kernel-include-regex "k_probe|k_threshold|k_correct|k_single" --output-format csv -d /tmp/prof_pmc -- python $REPO/bench.py --config $c --no-extras --steps 1 --warmup 0 > /dev/null 2> /tmp/prof_pmc.err)
    python tools/prof_summary.py pmc /tmp/prof_pmc | grep -v "^kernel" | grep "k_probe\|k_thresh\|k_correct\|k_single" >> $OUT/r3_config${c}_bench_pmc.txt
  done
done
python - "$OUT" $CONFIGS <<'PY'
import json, os, re, sys
sys.path.insert(0, os.getcwd())
import bench
out_dir, configs = sys.argv[1], sys.argv[2:]
path = os.path.join(out_dir, "r3_traffic.json")
doc = json.load(open("profiles/r3_traffic.json")) if os.path.exists("profiles/r3_traffic.json") else {"configs": {}}
if doc.get("sources") != bench.source_hashes():
    doc = {"configs": {}}      # passes of other sources do not mix with these
doc["sources"] = bench.source_hashes()
doc["note"] = ("per launch, mean over the launches of one step; `rocprofv3 --pmc <group> --kernel-include-regex 'k_probe|k_threshold|k_correct|k_single' -- "
               "python bench.py --config i --no-extras --steps 1 --warmup 0` (tools/measure_r3.sh; summaries in profiles/r3_config<i>_bench_pmc.txt); "
               "traffic = 2 x FETCH_SIZE (KB x 1024): gfx950 tallies a 128-byte fabric request as 64 B")
for c in configs:
    rec = {"k_probe": {}, "k_correct": {}, "k_single": {}}
    names = {"FETCH_SIZE": "fetch_size_kb", "WRITE_SIZE": "write_size_kb", "TCC_EA0_RDREQ_sum": "ea_rdreq", "TCC_HIT_sum": "tcc_hit", "TCC_MISS_sum": "tcc_miss",
             "SQ_INSTS_VALU": "insts_valu", "SQ_INSTS_SALU": "insts_salu", "SQ_INSTS_VMEM_RD": "insts_vmem_rd", "SQ_INSTS_LDS": "insts_lds",
             "SQ_ACTIVE_INST_VALU": "active_valu_quadcycles", "SQ_ACTIVE_INST_SCA": "active_scalar_quadcycles", "SQ_WAVE_CYCLES": "wave_quadcycles",
             "SQ_WAIT_ANY": "wait_any_quadcycles", "SQ_WAIT_INST_ANY": "wait_inst_any_quadcycles", "SQ_BUSY_CU_CYCLES": "busy_cu_cycles"}
    for line in open(os.path.join(out_dir, "r3_config%s_bench_pmc.txt" % c)):
        m = re.match(r"(k_\w+)(<.*>)?\s+(\S+)\s+(\d+)\s+(\S+)\s+(\S+)\s*$", line.rstrip())
        if not m or m.group(3) not in names:
            continue
        kern = m.group(1) if m.group(1) in ("k_correct", "k_single") else ("k_probe" if m.group(1) in ("k_probe_threshold_list", "k_probe_list", "k_probe") else None)
        if kern is None:
            continue
        n_disp, total = int(m.group(4)), float(m.group(6))
        rec[kern][names[m.group(3)]] = total / n_disp      # mean per launch
        rec[kern]["kernel"] = m.group(1) + (m.group(2) or "")
    doc["configs"][str(c)] = rec
json.dump(doc, open(path, "w"), indent=1)
print({c: (doc["configs"][c]["k_probe"].get("fetch_size_kb"), doc["configs"][c]["k_correct"].get("insts_valu")) for c in doc["configs"]})
PY
cp $OUT/r3_traffic.json profiles/r3_traffic.json   # (on the box: so that the bench lines below report it)
# host I/O microbenchmarks behind the end-to-end figures (tools/mb): page-cache reads / writes by method and thread count
(cd tools/mb && g++ -O2 -std=c++17 -pthread -o iob iob.cpp && g++ -O2 -std=c++17 -pthread -o iob2 iob2.cpp)
{ echo "# $(nproc) cores, $(grep MemTotal /proc/meminfo), /tmp on: $(df -T /tmp | tail -1)"; echo "## tools/mb/iob /tmp 4"; timeout 600 tools/mb/iob /tmp 4; echo "## tools/mb/iob /dev/shm 4"; timeout 600 tools/mb/iob /dev/shm 4; echo "## tools/mb/iob2 /tmp 4"; timeout 600 tools/mb/iob2 /tmp 4; } > $OUT/r3_host_io_microbench.txt 2>&1
rm -f /tmp/iob.dat /dev/shm/iob.dat
for c in $CONFIGS; do
  timeout 1800 python bench.py --config $c > $OUT/r3_bench_config$c.json 2> $OUT/r3_bench_config$c.err
  python tools/fmt_bench.py $OUT/r3_bench_config$c.json
done
