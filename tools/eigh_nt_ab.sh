#!/bin/bash
# round 4: streaming (nt) stores in the eigensolver's round kernel: tiles (1), rotations (2), both (3) -- C4 per generation
cd "$(dirname "$0")/.."
out=gpurun_out/eigh_nt_ab.txt; mkdir -p gpurun_out; : > $out
for rep in 1 2; do
  for lib in stochopy_amd/lib/libstochopy_hip.so build_ab/libsx_eigh_nt1.so build_ab/libsx_eigh_nt2.so build_ab/libsx_eigh_nt3.so; do
    echo "== $lib" >> $out
    python tools/ab_lib.py $lib tools/bench_c4.py 10 60 2>&1 | grep -v amdgpu.ids | head -1 >> $out
  done
done
cat $out
