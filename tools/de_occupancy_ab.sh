#!/bin/bash
cd "$(dirname "$0")/.."
python tools/de_occupancy_ab.py 2>&1 | grep -v amdgpu.ids
python tools/ab_lib.py build_ab/lib_w6/libstochopy_hip.so tools/de_occupancy_ab.py 2>&1 | grep -v amdgpu.ids
