#!/bin/bash
# usage: tools/prof.sh <tag> <bench args...>   -> gpurun_out/prof_<tag>/ (kernel trace + stats, csv)
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $OUT/bench.log 2>&1 < /dev/null
echo "rocprof rc=$?"
ls $OUT
for f in $(find $OUT -name "*kernel_stats.csv"); do echo "== $f"; head -12 "$f"; done
tail -1 $OUT/bench.log | cut -c1-700
