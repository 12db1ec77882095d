#!/bin/bash
cd "$(dirname "$0")/.."
for i in 1 2; do python tools/time_vd_minimize.py 16384 1024 2>&1 | grep -v amdgpu.ids; done
python tools/time_vd_calls.py 16384 1024 2>&1 | grep -v amdgpu.ids
python tools/time_vd_calls.py 16384 1024 2>&1 | grep -v amdgpu.ids
