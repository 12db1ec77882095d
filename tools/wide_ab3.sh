#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/wide_ab3.txt; mkdir -p gpurun_out; : > $O
echo "== parity (tests/test_gpu_wide.py) on the working tree's library (padded stage)" >> $O
timeout 900 python -m pytest tests/test_gpu_wide.py -x -q 2>&1 | tail -3 >> $O
echo "== working tree (SX_WIDE_PAD=1)" >> $O
timeout 900 python tools/bench_wide.py eval de pso vdcma >> $O 2>&1
echo "== build_ab/nopad (SX_WIDE_PAD=0)" >> $O
timeout 900 python tools/ab_lib.py build_ab/nopad/libstochopy_hip.so tools/bench_wide.py eval de pso vdcma >> $O 2>&1
echo "== counters, sx_eval rosenbrock n=4096 P=32768, padded" >> $O
bash tools/pmc_cmd.sh wide_pad $GRAFT_REPO_ROOT/tools/wide_one.py rosenbrock 4096 32768 >> $O 2>&1
echo "== counters, the same, unpadded" >> $O
bash tools/pmc_cmd.sh wide_nopad $GRAFT_REPO_ROOT/tools/ab_lib.py $GRAFT_REPO_ROOT/build_ab/nopad/libstochopy_hip.so $GRAFT_REPO_ROOT/tools/wide_one.py rosenbrock 4096 32768 >> $O 2>&1
cat $O
