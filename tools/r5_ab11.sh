#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab11.txt; : > $O
timeout 1200 python -m pytest tests/test_gpu_de.py -x -q -k "objectives" 2>&1 | tail -3 >> $O
echo "== working tree (eval_r8_long_kernel for rows of 257 ... 2048 elements off the grid, P >= 32768)" >> $O
timeout 600 python tools/eval_mid.py 2>&1 | grep -v amdgpu.ids >> $O
echo "== SX_EVAL_R8LONG=0" >> $O
SX_EVAL_R8LONG=0 timeout 600 python tools/eval_mid.py 2>&1 | grep -v amdgpu.ids >> $O
cat $O
