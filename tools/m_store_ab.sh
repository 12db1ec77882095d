#!/bin/bash
# round 4: store flavours / allocation flavours at the metric shape, alternating on one box
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/m_store_ab.txt; : > $out
for rep in 1 2; do
  for lib in stochopy_amd/lib/libstochopy_hip.so build_ab/libsx_st1.so build_ab/libsx_st2.so build_ab/libsx_st3.so; do
    python tools/ab_lib.py $lib tools/m_store_ab.py >> $out 2>&1
  done
  SX_POP_FLAGS=3 python tools/ab_lib.py stochopy_amd/lib/libstochopy_hip.so tools/m_store_ab.py >> $out 2>&1
  SX_POP_FLAGS=1 python tools/ab_lib.py stochopy_amd/lib/libstochopy_hip.so tools/m_store_ab.py >> $out 2>&1
  SX_POP_FLAGS=3 python tools/ab_lib.py build_ab/libsx_st1.so tools/m_store_ab.py >> $out 2>&1
done
cat $out
