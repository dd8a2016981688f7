#!/bin/bash
# VALU / SALU pipe utilisation of the bench kernels (one rocprofv3 --pmc pass per counter group)
REPO=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
for C in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  rm -rf /tmp/prof_pmc
  timeout 400 rocprofv3 --pmc $C --output-format csv -d /tmp/prof_pmc -- python $REPO/bench.py --cpu-sample 0 --steps 2 --warmup 0 "$@" > /dev/null 2> /tmp/prof_pmc.err
  python $REPO/tools/prof_summary.py pmc /tmp/prof_pmc | grep -v "^kernel" | grep "k_correct\|k_thresh\|k_probe"
done
