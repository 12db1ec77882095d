"""Per-generation timeline from a rocprofv3 kernel trace CSV: start (relative), duration, stream / queue of every kernel of one
generation in the middle of the run."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, nm in enumerate(names) if "wide_vd_candidates_kernel" in nm]
k = idx[len(idx) // 2]
k2 = idx[len(idx) // 2 + 1]
t0 = int(rows[k]["Start_Timestamp"])
for r in rows[k - 3:k2 + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s / 1e3:9.1f} {e / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  q={r.get('Queue_Id', '?'):>3} {r['Kernel_Name'][:80]}")
print("generation:", (int(rows[k2]["Start_Timestamp"]) - t0) / 1e3, "us")
