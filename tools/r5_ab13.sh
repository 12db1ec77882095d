#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab13.txt; : > $O
echo "== build_ab/dppnext (the neighbour of a Rosenbrock term by a DPP move, lane 7 loads it)" >> $O
timeout 900 python tools/pytest_with_lib.py build_ab/dppnext/libstochopy_hip.so tests/test_gpu_de.py -x -q -k "objectives" 2>&1 | tail -2 >> $O
for a in short mid; do
timeout 600 python tools/ab_lib.py build_ab/dppnext/libstochopy_hip.so tools/eval_mid.py $a 2>&1 | grep "rosenbrock" >> $O
done
echo "== the library in the tree (every lane loads its neighbour)" >> $O
for a in short mid; do
timeout 600 python tools/eval_mid.py $a 2>&1 | grep "rosenbrock" >> $O
done
cat $O
