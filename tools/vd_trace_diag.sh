#!/bin/bash
# a long wide VD-CMA run: the longest single kernel instances and the largest gaps on the GPU timeline
cd "$(dirname "$0")/.."
PROF_LINES=3 LOG_LINES=1 bash tools/prof_cmd.sh r5_vd200 $PWD/tools/run_vd_wide.py 16384 1024 200 > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_r5_vd200/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gen = 0
for r in rows:
    if "wide_vd_candidates" in r["Kernel_Name"]:
        gen += 1
    r["gen"] = gen
    r["dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("longest kernel instances:")
for r in sorted(rows, key=lambda r: -r["dur"])[:8]:
    print(f"   {r['dur']:10.1f} us  generation {r['gen']:4d}  {r['Kernel_Name'][:70]}")
gaps = []
for a, b in zip(rows, rows[1:]):
    gaps.append(((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3, a["gen"], a["Kernel_Name"][:40], b["Kernel_Name"][:40]))
print("largest gaps between consecutive kernels:")
for g in sorted(gaps, key=lambda g: -g[0])[:8]:
    print(f"   {g[0]:10.1f} us  after generation {g[1]:4d}: {g[2]} -> {g[3]}")
PY
