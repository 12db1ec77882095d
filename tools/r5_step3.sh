#!/bin/bash
# round 5: n + 8 doubles of LDS per row up to 256 elements (DE / PSO / eval kernels), single batch for short DE rows
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_de.py tests/test_gpu_pso.py tests/test_gpu_edges.py tests/test_gpu_immediate.py tests/test_gpu_external.py tests/test_gpu_configs_philox.py tests/test_distributed.py -q -x -m gpu 2>&1 | tail -4
python tools/de_occupancy_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/de_occupancy.txt
python tools/bench_shapes.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/shapes_step3.txt
python tools/bench_eval.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/eval_step3.txt
