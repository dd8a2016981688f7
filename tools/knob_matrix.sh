#!/bin/bash
# The GPU parity suites (C ABI, CLI goldens, fuzz) once per alternative code path of the library:
# slot layout, load factor, unfused / list-driven probe kernels on small batches, wave-per-read
# threshold kernel, no classification.  Run on the GPU box from the repository root.
cd "$(dirname "$0")/.."
for e in "RC_TABLE_LAYOUT=wide" "RC_TABLE_LOAD=0.85" "RC_TABLE_LOAD=0.25" "RC_LOCALITY=force" "RC_NO_FUSE=1 RC_LOCALITY=force" \
         "RC_NO_FUSE=1" "RC_K2_WAVE_PER_READ=1" "RC_NO_CLASSIFY=1" "RC_NO_ALT=1" "RC_K3_GENERIC=1" "RC_TABLE_FILTER=force RC_TABLE_FILTER_KIND=plain" "RC_TABLE_FILTER=force RC_TABLE_FILTER_KIND=core" "RC_TABLE_FILTER=search" "RC_NO_SINGLE=1" "RC_NO_BS_EXT=1"; do
  echo "== $e"
  env $e timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden_gpu.py tests/test_gpu_fuzz.py -m gpu -x -q \
      -k "not packed_and_wide" 2>&1 | grep -v "^Extension" | tail -2
done
