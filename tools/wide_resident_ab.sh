#!/bin/bash
# Which wide rows stay RESIDENT in LDS (the trial / new position kept there until the comparison) and which are STREAMED through the
# 4096-element stage: tools/bench_wide.py under SX_WIDE_RESIDENT_KB.  Output: gpurun_out/wide_resident_ab.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/wide_resident_ab.txt; : > $O
for kb in 148 72 40 20; do
  echo "== SX_WIDE_RESIDENT_KB=$kb" >> $O
  SX_WIDE_RESIDENT_KB=$kb timeout 900 python tools/bench_wide.py de pso de16 pso16 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O
