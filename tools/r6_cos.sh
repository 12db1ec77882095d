#!/bin/bash
# Round 6, VERDICT r5 item 4: the cosine objectives.  A/B of sx_eval (tools/bench_eval.py) and of C2 / C3a / C3b (tools/nt_ab.py):
#   libsx_eval_r5   = cos_mid(2 pi x) (rounds 4-5) + batches of eight loads in the eight-lanes-per-row kernel
#   libsx_eval_cos1 = cos_mid(2 pi x)              + batches of sixteen
#   libsx_eval_hb8  = cos_2pi(x) (round 6)         + batches of eight
#   product library = cos_2pi(x)                   + batches of sixteen
# build (here): see the hipcc lines in profiles/r6_cos.txt; run (GPU box): tools/r6_cos.sh
cd "$(dirname "$0")/.."
for rep in 1 2; do for lib in build_ab/libsx_eval_r5.so build_ab/libsx_eval_cos1.so build_ab/libsx_eval_hb8.so stochopy_amd/lib/libstochopy_hip.so; do
  echo "== $lib"; python tools/ab_lib.py $lib tools/bench_eval.py 2>&1 | grep -E "rastrigin|ackley|rosenbrock  n=  128 P= 1048576"
done; done
