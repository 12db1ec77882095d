"""Average duration per kernel name from a rocprofv3 kernel trace CSV (the first 5 calls of each left out)."""
import csv
import sys
from collections import defaultdict

d = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    w = v[5:] if len(v) > 10 else v
    if sum(v) > 20000:
        print(f"{sum(w) / len(w) / 1e3:9.1f} us x{len(v):4d}  {k[:90]}")
