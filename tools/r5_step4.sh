#!/bin/bash
# round 5: one-batch forms of the whole-wave DE / PSO kernels (rows of 129 ... 256 elements off the grid)
cd "$(dirname "$0")/.."
python -m pytest tests/test_gpu_de.py tests/test_gpu_pso.py tests/test_gpu_edges.py tests/test_gpu_configs_philox.py tests/test_gpu_external.py -q -x 2>&1 | tail -3
python tools/bench_shapes.py de pso 2>&1 | grep -v amdgpu.ids
python tools/de_occupancy_ab.py 2>&1 | grep -v amdgpu.ids | tail -6
