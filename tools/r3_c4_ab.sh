#!/bin/bash
# C4 after the round-3 changes to the CMA-ES generation's small kernels and the host's lagged look:
# tests of the CMA / eigensolver paths, bench_c4 with the look lagged (default) and not (SX_CMA_LAG=0), kernel stats.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/c4ab
mkdir -p $OUT
cd $R
python -m pytest tests/test_gpu_cmaes.py tests/test_gpu_eigh.py tests/test_gpu_vdcma.py -x -q 2>&1 | tail -4 | tee $OUT/pytest.txt
SX_CMA_LAG=1 python tools/bench_c4.py 2>&1 | grep -v amdgpu.ids | tee $OUT/c4_lag1.txt
SX_CMA_LAG=0 python tools/bench_c4.py 2>&1 | grep -v amdgpu.ids | tee $OUT/c4_lag0.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o run -- python $R/tools/run_c4.py 60 > $OUT/prof.log 2>&1 < /dev/null
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do cp $f $OUT/cmaes_c4_kernel_stats.csv; done
rm -rf $OUT/prof
head -25 $OUT/cmaes_c4_kernel_stats.csv | cut -c1-60,150-260
