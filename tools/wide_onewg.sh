#!/bin/bash
# A resident wide row whose LDS leaves one workgroup per CU (n ~ 9 000 ... 18 000): 512 threads + prefetch / 1024 threads /
# 1024 threads + prefetch / streamed instead (SX_WIDE_RESIDENT_KB=72).  Output: gpurun_out/wide_onewg.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/wide_onewg.txt; : > $O
for form in pre big bigpre; do
  echo "== SX_WIDE_ONE_WG=$form" >> $O
  SX_WIDE_ONE_WG=$form timeout 600 python tools/bench_wide.py de16 pso16 2>&1 | grep -v amdgpu.ids >> $O
done
echo "== SX_WIDE_RESIDENT_KB=72 (streamed)" >> $O
SX_WIDE_RESIDENT_KB=72 timeout 600 python tools/bench_wide.py de16 pso16 2>&1 | grep -v amdgpu.ids >> $O
cat $O
