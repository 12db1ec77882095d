#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab10.txt; : > $O
echo "== default (eval_r8_kernel: LDS-staged, compile-time n; Sphere: the one-visit kernel)" >> $O
timeout 600 python tools/eval_mid.py grid 2>&1 | grep -v amdgpu.ids >> $O
echo "== SX_EVAL_R8RT=2 (eval_r8_rt_kernel: straight from memory, run-time n)" >> $O
SX_EVAL_R8RT=2 timeout 600 python tools/eval_mid.py grid 2>&1 | grep -v amdgpu.ids >> $O
cat $O
