#!/bin/bash
# GPU box: MFMA counters of the CMA-ES contractions (C4: n=512, P=1024) -> gpurun_out/pmc_cmaes/mfma/
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_cmaes
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/cma_run.py <<'P'
import sys
sys.path.insert(0, "/root/repo")
import stochopy_amd as sa
o = {"popsize": 1024, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "eigh": "device", "maxiter": 12}
r = sa.optimize.minimize(sa.factory.rosenbrock, [[-5.12, 5.12]] * 512, method="cmaes", options=o)
print(r.nit, r.fun)
P
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES SQ_INSTS_VALU --output-format csv -d $OUT/mfma -o run -- python /tmp/cma_run.py > $OUT/mfma.log 2>&1 < /dev/null
echo "mfma rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python /tmp/cma_run.py > $OUT/stats.log 2>&1 < /dev/null
echo "stats rc=$?"
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT 2>&1 | grep -A8 "cma_gemm" | head -40
for f in $(find $OUT/stats -name "*kernel_stats.csv"); do grep -i "cma_\|Name" $f | cut -c1-200; done
