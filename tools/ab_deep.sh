#!/bin/bash
# A/B of the deep refinement step's entry rules: library variants built into stochopy_amd/lib_X (make OUTDIR=../lib_X
# EXTRA_CXXFLAGS=...), swapped in one after the other on the GPU box: per-sweep record, C4 time, eigensolver tests.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cp stochopy_amd/lib/libstochopy_hip.so /tmp/lib_default.so
for v in default "$@"; do
  echo "================ variant $v"
  if [ "$v" = default ]; then cp /tmp/lib_default.so stochopy_amd/lib/libstochopy_hip.so; else cp stochopy_amd/lib_$v/libstochopy_hip.so stochopy_amd/lib/libstochopy_hip.so; fi
  python tools/eigh_c4_kmax.py 512 1024 14 2>&1 | grep -v amdgpu.ids | grep -A3 "^gen  *\(2\|8\|11\|12\|14\) "
  for rep in 1 2; do python tools/bench_c4.py 10 60 2>&1 | grep 'device-resident loop:' | head -1; done
  timeout 900 python -m pytest tests/test_gpu_eigh.py -q -m gpu -n 4 2>&1 | tail -2
done
cp /tmp/lib_default.so stochopy_amd/lib/libstochopy_hip.so
