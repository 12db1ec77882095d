"""Per-call durations of the CPSO restart kernels from a rocprofv3 kernel trace: python tools/sel_hist.py <dir with *kernel_trace.csv>
(prints, per kernel, a histogram of durations and how the long calls of the selection relate to the generations that restart)"""
import csv, glob, sys
import numpy as np

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = ["pso_restart_select_kernel", "pso_radius_kernel", "select_finalize_kernel", "pso_generation_kernel", "pso_restart_apply_kernel"]
dur = {k: [] for k in names}
seq = []
for r in rows:
    for k in names:
        if k in r["Kernel_Name"]:
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            dur[k].append(d)
            seq.append((k, d, r["Kernel_Name"]))
for k in names:
    a = np.array(dur[k])
    if not len(a):
        continue
    h, edges = np.histogram(a, bins=[0, 2, 3, 4, 5, 6, 8, 10, 12, 15, 20, 25, 30, 40, 100])
    print(f"{k}: {len(a)} calls, mean {a.mean():.2f} us, median {np.median(a):.2f}")
    print("   " + "  ".join(f"<{edges[i+1]:g}:{h[i]}" for i in range(len(h)) if h[i]))
# a selection followed by an apply kernel (or a re-seeding generation kernel) = a generation that restarts
sel_then = {"restart": [], "plain": []}
for i, (k, d, full) in enumerate(seq):
    if k != "pso_restart_select_kernel":
        continue
    nxt = next((s for s in seq[i + 1 : i + 4] if s[0] in ("pso_generation_kernel", "pso_restart_apply_kernel")), None)
    restart = nxt is not None and (nxt[0] == "pso_restart_apply_kernel" or "true, true" in nxt[2].split("<")[-1])
    sel_then["restart" if restart else "plain"].append(d)
for k, v in sel_then.items():
    if v:
        v = np.array(v)
        print(f"selection in generations that {k}: {len(v)} calls, mean {v.mean():.2f} us, min {v.min():.2f}, max {v.max():.2f}")
sel = np.array(dur["pso_restart_select_kernel"])
print("selection durations, every 40th call:", " ".join(f"{x:.1f}" for x in sel[::40]))
