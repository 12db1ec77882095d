"""Per-call durations of cpso_post_kernel from a rocprofv3 kernel trace of tools/run_cpso_c3b.py: histogram, and the generation
numbers of the longest calls.  usage: post_hist.py <dir with *kernel_trace.csv>"""
import csv, glob, sys
import numpy as np

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(path)) if "cpso_post_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows])
h, e = np.histogram(d, bins=[0, 6, 7, 8, 10, 12, 15, 18, 21, 24, 27, 30, 35, 40, 50, 100, 1000])
print(f"cpso_post_kernel: {len(d)} calls, mean {d.mean():.2f} us, median {np.median(d):.2f}, max {d.max():.1f}")
print("   " + "  ".join(f"<{e[i+1]:g}:{h[i]}" for i in range(len(h)) if h[i]))
top = np.argsort(-d)[:12]
print("   longest calls (call number: us): " + ", ".join(f"{i}: {d[i]:.1f}" for i in sorted(top)))
