#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/wide_ab6.txt; mkdir -p gpurun_out; : > $O
echo "== parity on the working tree's library" >> $O
timeout 1200 python -m pytest tests/test_gpu_wide.py tests/test_gpu_vdcma.py -x -q 2>&1 | tail -3 >> $O
echo "== working tree" >> $O
timeout 900 python tools/bench_wide.py eval de pso vdcma >> $O 2>&1
echo "== HEAD (build_ab/base)" >> $O
timeout 900 python tools/ab_lib.py build_ab/base/libstochopy_hip.so tools/bench_wide.py de pso vdcma >> $O 2>&1
cat $O
