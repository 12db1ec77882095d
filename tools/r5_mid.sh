#!/bin/bash
# round 5, mid-round check: the whole GPU suite, smoke, the driver-style bench line, the shapes table
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5
python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r5/pytest_gpu_mid.txt; cat gpurun_out/r5/pytest_gpu_mid.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5/bench_N1_driver_args_mid.json 2> gpurun_out/r5/bench_err_mid.txt; tail -c 300 gpurun_out/r5/bench_N1_driver_args_mid.json; tail -3 gpurun_out/r5/bench_err_mid.txt
