#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_calib
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o run -- python $GRAFT_REPO_ROOT/tools/calib.py > $OUT/log.txt 2>&1 < /dev/null
echo rc=$?; grep calib $OUT/log.txt
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/fetch/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "eval_kernel" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:40], r["Grid_Size"])].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, "FETCH_SIZE mean", sum(v)/len(v), "n", len(v))
PY
