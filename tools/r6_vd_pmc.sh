#!/bin/bash
# PMC passes (separate, counters only) over a short wide VD-CMA run: HBM traffic and issue statistics of the candidates kernel.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6vd
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pmc() { tag=$1; ctr=$2; shift 2
  timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$tag -o run -- "$@" > $OUT/pmc_$tag.log 2>&1 < /dev/null
  echo "pmc $tag rc=$?"; }
pmc fetch FETCH_SIZE python $R/tools/trace_vd_wide.py 16384 1024 20
pmc write WRITE_SIZE python $R/tools/trace_vd_wide.py 16384 1024 20
pmc sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" python $R/tools/trace_vd_wide.py 16384 1024 20
pmc sq2 "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" python $R/tools/trace_vd_wide.py 16384 1024 20
python - <<PY
import csv, glob, json
from collections import defaultdict
out = {}
for tag in ("fetch", "write", "sq", "sq2"):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        for c, v in d.items():
            out.setdefault(k, {})[c] = [sum(v) / len(v), len(v)]
json.dump(out, open("$OUT/vd_pmc.json", "w"), indent=1)
for k, d in out.items():
    if "candidates" in k or "moments_partial" in k or "chain" in k:
        print(k)
        for c, (m, cnt) in sorted(d.items()):
            print("   %-28s %16.1f  (%d dispatches)" % (c, m, cnt))
PY
rm -rf $OUT/pmc_*/
