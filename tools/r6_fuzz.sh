#!/bin/bash
# round 6: the randomised-parity tools on the current build + the 2-rank bench rehearsal on one GPU -> gpurun_out/r6_fuzz.txt
cd "$(dirname "$0")/.."
out=gpurun_out/r6_fuzz.txt; mkdir -p gpurun_out; : > $out
{ echo "fuzz_deferred (philox):"; python tools/fuzz_deferred.py 2>&1 | tail -1
  echo "fuzz_deferred (RNG=numpy-legacy):"; RNG=numpy-legacy python tools/fuzz_deferred.py 2>&1 | tail -1
  echo "fuzz_immediate:"; python tools/fuzz_immediate.py 2>&1 | tail -1
  echo "fuzz_cpso_graph:"; python tools/fuzz_cpso_graph.py 90 4 2>&1 | tail -1
  echo "fuzz_round2 (60 s per family, FUZZ_SEED=${FUZZ_SEED:-61}):"; FUZZ_SEED=${FUZZ_SEED:-61} python tools/fuzz_round2.py 60 2>&1 | tail -8
  echo "fuzz_sharded:"; python tools/fuzz_sharded.py 2>&1 | tail -2
  echo "bench.py --gpus 2 on one GPU (gloo process group, both ranks on device 0):"; bash tools/bench_two_ranks_one_gpu.sh
} 2>&1 | grep -v amdgpu.ids >> $out
cat $out
