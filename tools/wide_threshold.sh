#!/bin/bash
cd "$(dirname "$0")/.."
python tools/wide_threshold.py 2>&1 | grep -v amdgpu.ids
python tools/ab_lib.py build_ab/lib2048/libstochopy_hip.so tools/wide_threshold.py 2>&1 | grep -v amdgpu.ids
python tools/pytest_with_lib.py build_ab/lib2048/libstochopy_hip.so tests/test_gpu_wide.py tests/test_gpu_de.py tests/test_gpu_pso.py tests/test_gpu_edges.py tests/test_gpu_external.py tests/test_gpu_vdcma.py -q -x 2>&1 | tail -4
