"""Per-dispatch view of the eigensolver's round kernel from a rocprofv3 kernel trace (csv):
duration of the launches that sweep, duration of the no-op launches after convergence, and the gap between
consecutive launches.  usage: eigh_gaps.py <dir with *kernel_trace.csv>"""
import csv, glob, sys
import numpy as np
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)))))
rows.sort()
prev_end, data = None, {}
for s, e, name, grid in rows:
    if "eigh_round_kernel" in name:
        key = grid
        d = data.setdefault(key, {"dur": [], "gap": []})
        d["dur"].append(e - s)
        if prev_end is not None and prev_name:
            d["gap"].append(s - prev_end)
    prev_end, prev_name = e, "eigh_round_kernel" in name
for grid, d in sorted(data.items()):
    dur, gap = np.array(d["dur"]), np.array(d["gap"])
    work = dur[dur > 3000]
    noop = dur[dur <= 3000]
    print(f"grid {grid:8d} threads: {len(dur):6d} launches; sweeping launches {len(work):6d}: mean {work.mean()/1e3:6.2f} us "
          f"median {np.median(work)/1e3:6.2f} p10 {np.percentile(work,10)/1e3:6.2f} p90 {np.percentile(work,90)/1e3:6.2f}; "
          f"no-op launches {len(noop):6d}: mean {noop.mean()/1e3 if len(noop) else 0:5.2f} us; gap to the previous round launch: "
          f"mean {gap.mean()/1e3:5.2f} median {np.median(gap)/1e3:5.2f} us")
