#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_other
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python $GRAFT_REPO_ROOT/tools/bench_other.py > $OUT/log.txt 2>&1 < /dev/null
echo rc=$?
grep -E "^C3|^C4" $OUT/log.txt
for f in $(find $OUT -name "*kernel_stats.csv"); do head -14 "$f" | cut -c1-200; done
