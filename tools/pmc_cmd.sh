#!/bin/bash
# usage: tools/pmc_cmd.sh <tag> <python script + args...> -> gpurun_out/pmc_<tag>/{sq,sq2,lds,fetch}/ (separate --pmc passes, no tracing domains)
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o run -- python $CMD > $OUT/$name.log 2>&1 < /dev/null
  echo "$name rc=$?"; }
CMD="$*"
run sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES
run sq2 SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
run fetch FETCH_SIZE
python - <<PY
import csv, glob, collections
for d in ("sq","sq2","lds","fetch"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
        for k, v in acc.items():
            if "wide" in k or "eval" in k or "generation" in k:
                print(d, k, {c: round(x / max(1, cnt[(k, c)])) for c, x in v.items()})
PY
