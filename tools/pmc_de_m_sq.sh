#!/bin/bash
# SQ counters of the DE generation kernel at the metric shape (n=128, P=4096): instructions per wave (= 2 rows)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_de_m; mkdir -p $OUT
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -o run -- python $R/bench.py --no-cpu-baseline --no-minimize-wall --steps 400 --warmup 50 > $OUT/a.log 2>&1 < /dev/null
python - <<PY
import collections, csv, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "de_generation" in row["Kernel_Name"]:
            agg[row["Kernel_Name"][40:100]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    w = sum(d["SQ_WAVES"]) / len(d["SQ_WAVES"])
    print(k, "waves", w, "  ".join(f"{c[3:]}={sum(v)/len(v)/w:.0f}" for c, v in sorted(d.items())))
PY
