#!/bin/bash
# usage: tools/pmc.sh <tag> <bench args...>  -> gpurun_out/pmc_<tag>/{sq,sq2,fetch,write}/  (separate --pmc passes, no tracing domains)
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o run -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --kernel-timing-launches 50 $BENCHARGS > $OUT/$name.log 2>&1 < /dev/null
  echo "$name rc=$?"
}
BENCHARGS="$*"
run sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES
run sq2 SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT
run fetch FETCH_SIZE
run write WRITE_SIZE
find $OUT -name "*counter_collection.csv" | head
