#!/bin/bash
# dev: the files-to-files run of tools/exp_e2e_variants.py under rocprofv3 --kernel-trace, several times: when a run's loop is
# slow (43 ms per 1 M-read batch instead of 11 on some boxes), are the kernels long or is the time between them?
# usage: tools/exp_e2e_trace.sh [runs]   (run on the GPU box from the repository root)
REPO=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-4}
export TMPDIR=/tmp
cd /tmp
python - <<PY
import sys, os
sys.path.insert(0, "$REPO"); sys.path.insert(0, "$REPO/tools")
import numpy as np, torch, synth_int
from exp_e2e_variants import write_fq
L, n = 150, 25_000_000
os.makedirs("/tmp/rc_e2e_tr", exist_ok=True)
gen = synth_int.Synth(1002, L, 30000, 1500, 0.8, 0.005, True, device=torch.device("cuda", 0))
seq, qual = gen.generate(0, n // 2)
S = seq.view(n, L + 1)[:, :L].cpu().numpy(); Q = qual.view(n, L + 1)[:, :L].cpu().numpy()
write_fq("/tmp/rc_e2e_tr/x_1.fq", S[:n // 2], Q[:n // 2], L); write_fq("/tmp/rc_e2e_tr/x_2.fq", S[n // 2:], Q[n // 2:], L)
PY
cd /tmp/rc_e2e_tr
for i in $(seq 1 $N); do
  rm -rf out prof; sync
  RC_TIMING=1 rocprofv3 --kernel-trace --stats --output-format csv -d prof -- $REPO/rcorrector_amd/rcorrector -p x_1.fq x_2.fq -k 23 -od out 2> err.txt > /dev/null
  grep -E "correction loop|stage totals" err.txt | cut -c1-220
  python $REPO/tools/prof_summary.py stats prof 2>/dev/null | grep -E "k_probe_threshold|k_correct|k_single|k_fix_list|copyBuffer" | head -6
done
rm -rf /tmp/rc_e2e_tr
