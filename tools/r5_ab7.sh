#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/wide_onewg.sh > /dev/null 2>&1
O=gpurun_out/eval_mid_ab.txt; : > $O
echo "== working tree (plain staging loop)" >> $O
timeout 600 python tools/eval_mid.py 2>&1 | grep -v amdgpu.ids >> $O
echo "== build_ab/noplain (-DSX_EVAL_PLAIN_LOOP=0)" >> $O
timeout 600 python tools/ab_lib.py build_ab/noplain/libstochopy_hip.so tools/eval_mid.py 2>&1 | grep -v amdgpu.ids >> $O
timeout 600 python -m pytest tests/test_gpu_wide.py tests/test_gpu_vdcma.py tests/test_gpu_edges.py -x -q 2>&1 | tail -2 > gpurun_out/ab7_pytest.txt
cat gpurun_out/wide_onewg.txt $O gpurun_out/ab7_pytest.txt
