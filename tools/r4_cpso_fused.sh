#!/bin/bash
# round 4: CPSO with the radius decision taken from the generation kernel's by-product (SX_CPSO_FUSED_RADIUS=0: the radius pass every generation)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4b
timeout 1500 python -m pytest tests/test_gpu_pso.py tests/test_gpu_configs_philox.py -q -x 2>&1 | tail -3
{ echo "fuzz_cpso_graph:"; timeout 600 python tools/fuzz_cpso_graph.py 60 4 2>&1 | tail -1
  for k in 1 2; do
    echo "== fused radius decision"; timeout 300 python tools/bench_cpso.py 2>&1 | grep -v amdgpu.ids | tail -3
    echo "== SX_CPSO_FUSED_RADIUS=0"; SX_CPSO_FUSED_RADIUS=0 timeout 300 python tools/bench_cpso.py 2>&1 | grep -v amdgpu.ids | tail -3
  done; } > gpurun_out/r4b/cpso_fused_radius.txt 2>&1
cat gpurun_out/r4b/cpso_fused_radius.txt
PROF_LINES=8 tools/prof_cmd.sh selhist $GRAFT_REPO_ROOT/tools/run_cpso_c3b.py 1200 > gpurun_out/r4b/selprof.log 2>&1
python tools/sel_hist.py gpurun_out/prof_selhist > gpurun_out/r4b/sel_hist.txt 2>&1
cat gpurun_out/r4b/sel_hist.txt
find gpurun_out/prof_selhist -name "*kernel_trace.csv" -delete
