#!/bin/bash
cd "$(dirname "$0")/.."
python tools/bench_wide.py eval de pso 2>&1 | grep -v amdgpu.ids
echo "== build_ab/lib_sl16 (SX_WIDE_STREAM_LEAVES=16)"
python tools/ab_lib.py build_ab/lib_sl16/libstochopy_hip.so tools/bench_wide.py eval de pso 2>&1 | grep -v amdgpu.ids
