#!/bin/bash
# rocprofv3 kernel stats of the updating="immediate" sweeps and of the propose/select path around a caller-supplied objective
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for t in immediate external; do
  OUT=$R/gpurun_out/prof_$t; mkdir -p $OUT
  if [ $t = immediate ]; then CMD="python $R/tools/bench_immediate.py de:rosenbrock:128:4096 pso:ackley:256:16384"; else CMD="python $R/tools/bench_external.py"; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- $CMD > $OUT/run.log 2>&1 < /dev/null
  echo "$t rc=$?"; grep -v amdgpu.ids $OUT/run.log | tail -9
  for f in $(find $OUT -name "*kernel_stats.csv"); do head -8 "$f" | cut -c1-260; done
done
