#!/bin/bash
# A/B of two builds of the library on one GPU box: stochopy_amd/lib_<variant>/libstochopy_hip.so swapped in for the product's,
# alternating.  usage: ab_swap.sh <variant> <python script + args>
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
v=$1; shift
cp stochopy_amd/lib/libstochopy_hip.so /tmp/lib_default.so
for rep in 1 2; do
  for w in default $v; do
    if [ $w = default ]; then cp /tmp/lib_default.so stochopy_amd/lib/libstochopy_hip.so; else cp stochopy_amd/lib_$v/libstochopy_hip.so stochopy_amd/lib/libstochopy_hip.so; fi
    echo "== $w"; python "$@" 2>&1 | grep -v amdgpu.ids
  done
done
cp /tmp/lib_default.so stochopy_amd/lib/libstochopy_hip.so
