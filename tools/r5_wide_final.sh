#!/bin/bash
# Round 5, after the wide kernels' second pass (kWideFrom = 2048): the whole GPU suite, the wide-row table, VD-CMA's candidates
# kernel with 256- and 512-thread workgroups, the off-grid shapes.  Output: gpurun_out/r5w/
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5w; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/pytest_gpu.txt
timeout 900 python tools/bench_wide.py eval de pso vdcma 2>&1 | grep -v amdgpu.ids > $O/wide_rows.txt
echo "== SX_VD_THREADS=512" > $O/vd_threads.txt
SX_VD_THREADS=512 timeout 600 python tools/bench_wide.py vdcma 2>&1 | grep -v amdgpu.ids >> $O/vd_threads.txt
echo "== SX_VD_THREADS=256" >> $O/vd_threads.txt
SX_VD_THREADS=256 timeout 600 python tools/bench_wide.py vdcma 2>&1 | grep -v amdgpu.ids >> $O/vd_threads.txt
timeout 900 python tools/bench_shapes.py de pso 2>&1 | grep -v amdgpu.ids > $O/shapes.txt
cat $O/pytest_gpu.txt $O/wide_rows.txt $O/vd_threads.txt
