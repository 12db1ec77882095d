#!/bin/bash
# round 4, final evidence in one GPU-box call: profiles (tools/r4_profiles.sh), the objective-only kernel, in-kernel stamps of the
# metric shape, the whole GPU suite
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4
bash tools/r4_profiles.sh > gpurun_out/r4/profiles.log 2>&1; tail -3 gpurun_out/r4/profiles.log
bash tools/bench_eval.sh > gpurun_out/r4/eval_kernel.txt 2>&1; cat gpurun_out/r4/eval_kernel.txt
python tools/trace.py 128 4096 2>&1 | grep -v amdgpu.ids | tail -14 > gpurun_out/r4/de_M_trace.txt; cat gpurun_out/r4/de_M_trace.txt
python -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/r4/pytest_gpu.txt; cat gpurun_out/r4/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
