#!/bin/bash
# round 5, first GPU-box call: the whole GPU suite on the wide-row tree, the driver-style bench line, wide rows, off-grid shapes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5
python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r5/pytest_gpu.txt; cat gpurun_out/r5/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5/bench_N1_driver_args.json 2> gpurun_out/r5/bench_err.txt; tail -c 600 gpurun_out/r5/bench_N1_driver_args.json
timeout 900 python tools/bench_shapes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5/shapes.txt; cat gpurun_out/r5/shapes.txt
timeout 900 python tools/bench_wide.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5/wide.txt; cat gpurun_out/r5/wide.txt
