#!/bin/bash
# usage: tools/prof_cmd.sh <tag> <python script + args...>  -> gpurun_out/prof_<tag>/ (rocprofv3 kernel trace + stats, csv)
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout ${PROF_TIMEOUT:-400} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python "$@" > $OUT/log.txt 2>&1 < /dev/null
echo "rocprof rc=$?"
for f in $(find $OUT -name "*kernel_stats.csv"); do echo "== $f"; head -${PROF_LINES:-16} "$f" | cut -c1-220; done
grep -v "^W2026\|^I2026\|^E2026" $OUT/log.txt | tail -${LOG_LINES:-8} | cut -c1-400
