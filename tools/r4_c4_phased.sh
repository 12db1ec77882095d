#!/bin/bash
# round 4: CMA-ES decompositions enqueued in pieces (SX_CMA_PHASED=0: a whole decomposition + one sweep of allowance ahead)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4b
timeout 1500 python -m pytest tests/test_gpu_cmaes.py tests/test_gpu_eigh.py -q -x 2>&1 | tail -3
{ for k in 1 2; do
    echo "== pieces"; timeout 300 python tools/bench_c4.py 2>&1 | grep "C4 cmaes"
    echo "== SX_CMA_PHASED=0"; SX_CMA_PHASED=0 timeout 300 python tools/bench_c4.py 2>&1 | grep "C4 cmaes"
  done; } > gpurun_out/r4b/c4_phased.txt 2>&1
cat gpurun_out/r4b/c4_phased.txt
