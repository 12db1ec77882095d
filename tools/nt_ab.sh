#!/bin/bash
# round 4: streaming stores (default build) vs plain stores (nt0) vs streaming stores + streaming loads (ntl1: own rows, ntl3: + donors)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/nt_ab.txt; : > $out
for rep in 1 2; do
  for lib in build_ab/libsx_nt0.so stochopy_amd/lib/libstochopy_hip.so build_ab/libsx_ntl1.so build_ab/libsx_ntl3.so; do
    python tools/ab_lib.py $lib tools/nt_ab.py 2>&1 | grep -v amdgpu.ids >> $out
  done
done
cat $out
