#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/wide_ab4.txt; mkdir -p gpurun_out; : > $O
echo "== parity (tests/test_gpu_wide.py) on the working tree's library" >> $O
timeout 900 python -m pytest tests/test_gpu_wide.py -x -q 2>&1 | tail -3 >> $O
echo "== working tree (per-piece shuffle finish, unpredicated full chunks / leaves)" >> $O
timeout 900 python tools/bench_wide.py eval de pso vdcma >> $O 2>&1
echo "== build_ab/fin0 (SX_WIDE_FINISH_SHFL=0: level-by-level finish in LDS)" >> $O
timeout 900 python tools/ab_lib.py build_ab/fin0/libstochopy_hip.so tools/bench_wide.py eval de pso >> $O 2>&1
echo "== HEAD (build_ab/base)" >> $O
timeout 600 python tools/ab_lib.py build_ab/base/libstochopy_hip.so tools/bench_wide.py eval >> $O 2>&1
echo "== counters, sx_eval rosenbrock n=4096 P=32768" >> $O
bash tools/pmc_cmd.sh wide_new $GRAFT_REPO_ROOT/tools/wide_one.py rosenbrock 4096 32768 >> $O 2>&1
cat $O
