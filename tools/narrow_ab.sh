#!/bin/bash
# The fused leaves reduction of the wavefront-per-row kernels (rows of 257 ... 2560 elements of any length): select-on-chain
# short leaves + tail by lane (working tree) against a build without them (build_ab/nofs).  Output: gpurun_out/narrow_ab.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/narrow_ab.txt; mkdir -p gpurun_out; : > $O
echo "== parity on the working tree's library (tests/test_gpu_de.py test_gpu_pso.py test_gpu_edges.py test_gpu_wide.py)" >> $O
timeout 1500 python -m pytest tests/test_gpu_de.py tests/test_gpu_pso.py tests/test_gpu_edges.py tests/test_gpu_wide.py -x -q 2>&1 | tail -3 >> $O
echo "== working tree" >> $O
timeout 900 python tools/bench_shapes.py de pso >> $O 2>&1
echo "== build_ab/nofs (-DSX_FUSED_SELECT=0 -DSX_FUSED_TAIL_BY_LANE=0)" >> $O
timeout 900 python tools/ab_lib.py build_ab/nofs/libstochopy_hip.so tools/bench_shapes.py de pso >> $O 2>&1
cat $O
