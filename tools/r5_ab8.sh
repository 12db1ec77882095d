#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab8.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_wide.py tests/test_gpu_vdcma.py tests/test_gpu_edges.py -x -q 2>&1 | tail -2 >> $O
echo "== working tree (resident rows <= 72 KB of LDS, generic DE kernel two elements at a time)" >> $O
timeout 900 python tools/bench_wide.py eval de pso de16 pso16 vdcma 2>&1 | grep -v amdgpu.ids >> $O
echo "== generic strategies / constraints" >> $O
timeout 600 python tools/bench_wide.py degen 2>&1 | grep -v amdgpu.ids >> $O
echo "== HEAD~2 (build_ab/base) generic strategies / constraints" >> $O
timeout 600 python tools/ab_lib.py build_ab/base/libstochopy_hip.so tools/bench_wide.py degen 2>&1 | grep -v amdgpu.ids >> $O
echo "== SX_VD_THREADS=384" >> $O
SX_VD_THREADS=384 timeout 600 python tools/bench_wide.py vdcma 2>&1 | grep -v amdgpu.ids >> $O
cat $O
