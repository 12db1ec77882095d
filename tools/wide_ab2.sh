#!/bin/bash
# A/B of the wide-row kernels on one box: the working tree's library against a build of HEAD (build_ab/base), and the
# resident-row limit (SX_WIDE_RESIDENT_KB).  Output: gpurun_out/wide_ab2.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/wide_ab2.txt; mkdir -p gpurun_out; : > $O
echo "== parity (tests/test_gpu_wide.py) on the working tree's library" >> $O
timeout 900 python -m pytest tests/test_gpu_wide.py -x -q 2>&1 | tail -3 >> $O
echo "== working tree" >> $O
timeout 600 python tools/bench_wide.py eval de >> $O 2>&1
echo "== HEAD (build_ab/base)" >> $O
timeout 600 python tools/ab_lib.py build_ab/base/libstochopy_hip.so tools/bench_wide.py eval de >> $O 2>&1
echo "== working tree, SX_WIDE_RESIDENT_KB=72" >> $O
SX_WIDE_RESIDENT_KB=72 timeout 600 python tools/bench_wide.py de >> $O 2>&1
cat $O
