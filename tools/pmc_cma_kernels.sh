#!/bin/bash
# SQ counters of the CMA-ES GEMM kernels at one shape (default n=512 P=1024); csv per pass under gpurun_out/pmc_cma/
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/pmc_cma; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for ctr in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $OUT/p$i -o run -- python $R/tools/bench_cma_kernels.py ${1:-512} ${2:-1024} > $OUT/p$i.log 2>&1 < /dev/null
  echo "pass $i rc=$?"
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        import re
        kn = r["Kernel_Name"]; m = re.search(r"cma_gemm_kernel<(\d+), (\d+), (\d+)", kn)
        k = ("gemm<%s,%s,%s>" % m.groups()) if m else kn.split("(")[0][-50:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] in ("SQ_WAVES", "SQ_WAIT_INST_LDS", "SQ_VALU_MFMA_BUSY_CYCLES"): cnt[(k, r["Counter_Name"])] += 1
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()):
        n = max(cnt.get((k, "SQ_WAVES"), 1), 1)
        print(f"   {c:28s} {x / n:14.1f} per launch")
PY
