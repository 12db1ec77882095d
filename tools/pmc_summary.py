"""Summarise rocprofv3 --pmc counter_collection CSVs: per-kernel mean of each counter per dispatch."""
import collections, csv, glob, sys
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:90]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    if "de_generation" not in k and "pso_generation" not in k and "cma_gemm" not in k and "select_finalize" not in k:
        continue
    print(k)
    for c, v in sorted(d.items()):
        v = sorted(v)
        print(f"   {c:26s} n={len(v):5d} mean={sum(v)/len(v):14.1f} median={v[len(v)//2]:14.1f}")
