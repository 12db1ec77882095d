#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab14.txt; : > $O
echo "== working tree (wave index scalar)" >> $O
timeout 900 python tools/bench_shapes.py de pso large 2>&1 | grep -v amdgpu.ids >> $O
echo "== build_ab/nouni (-DSX_UNIFORM_WAVE_INDEX=0)" >> $O
timeout 900 python tools/ab_lib.py build_ab/nouni/libstochopy_hip.so tools/bench_shapes.py de pso large 2>&1 | grep -v amdgpu.ids >> $O
cat $O
