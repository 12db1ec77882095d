#!/bin/bash
# A/B on one box: rows of exactly 256 elements through the fused-term reduction (SX_FUSED_ABOVE=255) vs the staged one
set -u
cd $GRAFT_REPO_ROOT
SRC=stochopy_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -DSX_FUSED_ABOVE=255 -shared -x hip $SRC/*.hip $SRC/*.cpp -o /tmp/libsx_fused255.so 2>&1 | grep -E "error" | head
for lib in stochopy_amd/lib/libstochopy_hip.so /tmp/libsx_fused255.so; do
  echo "== $lib"
  python tools/ab_lib.py $lib tools/bench_pso.py ackley:256:16384 sphere:256:16384 rosenbrock:256:16384
done
