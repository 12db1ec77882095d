#!/bin/bash
# round 5: run-time register-chain objective for one-batch rows + host pool: targeted tests, shapes, host-pool bench
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_de.py tests/test_gpu_pso.py tests/test_gpu_edges.py tests/test_gpu_immediate.py tests/test_gpu_external.py -q -x 2>&1 | tail -8 > gpurun_out/r5/pytest_step2.txt; cat gpurun_out/r5/pytest_step2.txt
timeout 600 python tools/bench_shapes.py de pso 2>&1 | grep -v amdgpu.ids > gpurun_out/r5/shapes_chain_rt.txt; cat gpurun_out/r5/shapes_chain_rt.txt
timeout 600 python tools/bench_host_pool.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5/host_pool.txt; cat gpurun_out/r5/host_pool.txt
