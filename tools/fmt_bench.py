import json,sys
for line in (open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin):
    if not line.startswith("{"): continue
    d=json.loads(line); c=d["config"]; print(c["workload"]); print("  reads/s=%.1fM step=%.1fms" % (d["value"]/1e6, d["ms_per_step"]), c["kernel_ms_per_step"], "frac=%.3f corrected=%.3f" % (d["roofline"]["frac"], c["reads_corrected_frac"]), "cpu:", (d["cpu_baseline"] or {}).get("value"), (d["cpu_baseline"] or {}).get("gpu_matches_oracle_on_sample"))
