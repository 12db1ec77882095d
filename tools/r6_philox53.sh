#!/bin/bash
# Round 6, VERDICT r5 item 3: what 53-bit Philox uniforms (DE crossover, PSO r1 / r2) cost against the 32-bit ones.
# Build here (hipcc):   tools/r6_philox53.sh build     -> build_ab/libsx_philox53.so (the fused narrow DE / PSO kernels with -DSX_PHILOX53=1)
# Measure on the GPU:   tools/r6_philox53.sh run       -> M, C2 (tools/nt_ab.py de) and C3a, C3b (pso), three alternating repetitions
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p build_ab
  F="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function -DSX_PHILOX53=1"
  PRE=$(echo 'int main(){}' | /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=8 -c - -o /dev/null 2>/dev/null && echo -mllvm -amdgpu-kernarg-preload-count=8)
  for f in sx_de sx_de_p2p sx_pso; do /opt/rocm/bin/hipcc $F -c stochopy_amd/csrc/$f.hip -o build_ab/${f}_p53.o & done
  /opt/rocm/bin/hipcc $F $PRE -c stochopy_amd/csrc/sx_de_chain.hip -o build_ab/sx_de_chain_p53.o &
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_ab/sx_de_p53.o build_ab/sx_de_chain_p53.o build_ab/sx_de_p2p_p53.o build_ab/sx_pso_p53.o \
      $(ls stochopy_amd/lib/*.o | grep -v -E "/(sx_de|sx_de_chain|sx_de_p2p|sx_pso)\.o") -o build_ab/libsx_philox53.so
  exit $?
fi
for rep in 1 2 3; do for lib in stochopy_amd/lib/libstochopy_hip.so build_ab/libsx_philox53.so; do
  python tools/ab_lib.py $lib tools/nt_ab.py de pso 2>&1 | grep -v amdgpu.ids
done; done
