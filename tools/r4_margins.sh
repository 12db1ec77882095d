#!/bin/bash
# round 4: whole-run margins of the device-resident CMA-ES loop (device eigensolver WITH the refinement step)
# against the oracle (LAPACK + canonical signs) over several seeds / shapes, and the eigensolver fuzz family with the step allowed
cd "$(dirname "$0")/.."
out=gpurun_out/r4_margins.txt; mkdir -p gpurun_out; : > $out
{ for seed in 0 1 2 3; do C4_BRIEF=1 python tools/c4_parity_margin.py 60 $seed 512 1024 rosenbrock; done
  C4_BRIEF=1 python tools/c4_parity_margin.py 60 5 566 1132 rosenbrock
  C4_BRIEF=1 python tools/c4_parity_margin.py 60 6 512 1024 rastrigin
  C4_BRIEF=1 python tools/c4_parity_margin.py 80 7 257 520 sphere
  echo "fuzz_round2 eigh family, SX_EIGH_REFINE=1 (limits of the refinement rule), 120 s:"
  SX_EIGH_REFINE=1 FUZZ_ONLY=eigh FUZZ_SEED=57 python tools/fuzz_round2.py 120
} 2>&1 | grep -v amdgpu.ids >> $out
cat $out
