#!/bin/bash
# round 4: kernarg preload of the chained DE kernel's leading arguments (pre8) vs the same source without the flag (nopre)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/pre_ab.txt; : > $out
for rep in 1 2 3; do
  for lib in build_ab/libsx_nopre.so build_ab/libsx_pre8.so; do
    python tools/ab_lib.py $lib tools/nt_ab.py de 2>&1 | grep -v amdgpu.ids >> $out
  done
done
cat $out
python -m pytest tests/test_gpu_de.py -x -q 2>&1 | tail -5
