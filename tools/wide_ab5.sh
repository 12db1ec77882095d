#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/wide_ab5.txt; mkdir -p gpurun_out; : > $O
echo "== parity on the working tree's library" >> $O
timeout 1200 python -m pytest tests/test_gpu_wide.py tests/test_gpu_vdcma.py -x -q 2>&1 | tail -3 >> $O
echo "== working tree (chunks by elements, tail by lane, select-on-chain short leaves, DE prefetch)" >> $O
timeout 900 python tools/bench_wide.py eval de pso vdcma >> $O 2>&1
echo "== build_ab/nopre (the same without the DE prefetch)" >> $O
timeout 900 python tools/ab_lib.py build_ab/nopre/libstochopy_hip.so tools/bench_wide.py de >> $O 2>&1
echo "== counters, sx_eval rosenbrock n=4096 P=32768" >> $O
bash tools/pmc_cmd.sh wide_new2 $GRAFT_REPO_ROOT/tools/wide_one.py rosenbrock 4096 32768 >> $O 2>&1
cat $O
