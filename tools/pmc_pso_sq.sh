#!/bin/bash
# SQ counters of the PSO generation kernel at C3 (Ackley and Sphere, n=256 P=16384): is it VALU- or memory-latency-bound?
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2/pmc_pso_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/pso_run.py <<'P'
import sys
sys.path.insert(0, "/root/repo")
import stochopy_amd as sa
for obj in ("ackley", "sphere"):
    o = {"popsize": 16384, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "maxiter": 60, "updating": "deferred"}
    r = sa.optimize.minimize(getattr(sa.factory, obj), [[-5.12, 5.12]] * 256, method="pso", options=o)
    print(obj, r.nit, r.fun)
P
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/a -o run -- python /tmp/pso_run.py > $OUT/a.log 2>&1 < /dev/null
echo "rc=$?"
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU --output-format csv -d $OUT/b -o run -- python /tmp/pso_run.py > $OUT/b.log 2>&1 < /dev/null
echo "rc=$?"
python - <<PY
import collections, csv, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "pso_generation" in row["Kernel_Name"]:
            agg[row["Kernel_Name"][:75]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:24s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
PY
