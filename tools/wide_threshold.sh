#!/bin/bash
# Where the one-workgroup-per-row kernels overtake the wavefront-per-row kernels: the same row lengths through the default
# build (kWideFrom = 2560) and through a build with -DSX_WIDE_FROM=1024 (build_ab/wf1024).  Output: gpurun_out/wide_threshold2.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/wide_threshold2.txt; mkdir -p gpurun_out; : > $O
NS="1025 1280 1536 1792 2048 2049 2304 2560"
python tools/wide_threshold.py $NS 2>&1 | grep -v amdgpu.ids >> $O
python tools/ab_lib.py build_ab/wf1024/libstochopy_hip.so tools/wide_threshold.py $NS 2>&1 | grep -v amdgpu.ids >> $O
cat $O
