#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab12.txt; : > $O
echo "== default (n = 512 / 1024 / 2048: compile-time plan, one wavefront per row; n > 2048: one workgroup per row)" >> $O
timeout 600 python tools/eval_mid.py long 2>&1 | grep -v amdgpu.ids >> $O
echo "== SX_EVAL_R8LONG=2 (eval_r8_long_kernel for all of them)" >> $O
SX_EVAL_R8LONG=2 timeout 600 python tools/eval_mid.py long 2>&1 | grep -v amdgpu.ids >> $O
cat $O
