#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/r2a/pytest.log
L=rcorrector_amd/librcorrector_amd.so
(
export BENCH_ARGS=""
tools/ab.sh $L rcorrector_amd/variants/w5.so rcorrector_amd/variants/heads1.so rcorrector_amd/variants/deq4.so
for g in 2 3 4 5; do echo "grid waves $g"; RC_K3_GRID_WAVES=$g tools/ab.sh $L; done
echo noclassify; RC_NO_CLASSIFY=1 tools/ab.sh $L
export BENCH_ARGS="--config 1"
tools/ab.sh $L rcorrector_amd/variants/w5.so
RC_PHASE_PROF=1 python bench.py --steps 1 --warmup 0 --cpu-sample 0 2>&1 | grep "phase prof"
python bench.py --steps 2 --warmup 1 2>&1 | tail -1
) > gpurun_out/r2a/ab.log 2>&1
