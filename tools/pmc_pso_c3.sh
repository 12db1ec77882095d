#!/bin/bash
# HBM traffic of the PSO (C3a) and CPSO (C3b) kernels, separate --pmc passes without tracing domains
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for which in pso:run_pso_c3.py:200 cpso:run_cpso_c3b.py:200; do
  tag=${which%%:*}; rest=${which#*:}; script=${rest%%:*}; m=${rest#*:}
  OUT=$R/gpurun_out/pmc_$tag; mkdir -p $OUT
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -o run -- python $R/tools/$script $m > $OUT/$c.log 2>&1 < /dev/null
    echo "$tag $c rc=$?"
  done
  python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("== $tag ($m generations)")
for k, d in agg.items():
    if "pso_" not in k and "finalize" not in k: continue
    print("  ", k)
    for c, v in sorted(d.items()):
        print(f"      {c:12s} launches={len(v):4d} mean={sum(v)/len(v):14.1f} KB")
PY
done
