#!/bin/bash
# HBM traffic of the PSO generation kernel at C3 (separate --pmc passes, no tracing domains)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_pso; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -o run -- python $R/tools/run_cpso_c3b.py > $OUT/$c.log 2>&1 < /dev/null
  echo "$c rc=$?"
done
python - <<'PY'
import csv, glob, collections, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_pso"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    if "pso_" not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:12s} launches={len(v):4d} mean={sum(v)/len(v):14.1f} KB")
PY
