#!/bin/bash
# dev: what the probe kernel's waves spend their cycles on -- rocprofv3 --pmc passes (8 SQ slots each) per library / environment.
# A pass that stops early (round 6: some boxes, now and then, right after the runtime comes up) is retried.
# Usage: tools/pmc_groups.sh "<label>=<env assignments or RC_LIB=...>" ...    (run on the GPU box from the repository root; CONFIG=2)
cd "$(dirname "$0")/.."
REPO=$(pwd)
export TMPDIR=/tmp
G1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE"
G2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
G3="SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT"
G4="TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
for spec in "$@"; do
  label=${spec%%=*}; envs=${spec#*=}
  echo "==== $label ($envs)"
  for G in "$G1" "$G2" "$G3" "$G4"; do
    for try in 1 2 3; do
      rm -rf /tmp/prof_pmc
      (cd /tmp && env $envs timeout ${PMC_TRY_S:-100} rocprofv3 --kernel-trace --pmc $G --kernel-include-regex "k_probe|k_correct|k_single" --output-format csv -d /tmp/prof_pmc -- python $REPO/bench.py --config ${CONFIG:-2} --cpu-sample 0 --no-extras --steps 2 --warmup 0 > /dev/null 2> /tmp/prof_pmc.err) && break
      echo "   (try $try of [$G] stopped: $(grep -v 'simple_timer\|output_stream' /tmp/prof_pmc.err | tail -1 | cut -c1-120))"
    done
    python tools/prof_summary.py pmc /tmp/prof_pmc | grep "k_probe_threshold" | awk '{printf "%-30s %16.0f (largest dispatch)\n", $(NF-3), $(NF-1)}'
    f=$(find /tmp/prof_pmc -name "*kernel_trace.csv" 2>/dev/null | head -1)
    [ -n "$f" ] && python - "$f" <<'PY'
import csv,sys
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"])) for r in csv.DictReader(open(sys.argv[1])) if "k_probe_threshold" in r["Kernel_Name"]]
if d: print("kernel_ms_largest             %16.3f" % (max(d)/1e6))
PY
  done
done
