#!/bin/bash
# round 6, final evidence in one GPU-box call: the whole GPU suite + smoke, then tools/r6_profiles.sh (bench lines, rocprofv3 kernel
# stats, PMC passes, the objective kernel, off-grid shapes, wide rows), the section 8(f) rows, the host pool
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6f
python -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/r6f/pytest_gpu.txt; cat gpurun_out/r6f/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r6f/smoke.txt
bash tools/r6_profiles.sh > gpurun_out/r6f/profiles.log 2>&1; tail -3 gpurun_out/r6f/profiles.log
python tools/r5_frows.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6f/frows_lines.txt; cat gpurun_out/r6f/frows_lines.txt
python tools/bench_host_pool.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6f/host_pool.txt; tail -12 gpurun_out/r6f/host_pool.txt
