cd $GRAFT_REPO_ROOT
for r in 1 2; do python tools/bench_c4.py 10 60 2>&1 | grep -v amdgpu | head -1; SX_CMA_TRI=0 python tools/bench_c4.py 10 60 2>&1 | grep -v amdgpu | head -1; done
python -m pytest tests/test_gpu_cmaes.py tests/test_gpu_vdcma.py -x -q 2>&1 | tail -3
python -m pytest tests/test_distributed.py -x -q -k "cmaes" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c4 -o run -- python $GRAFT_REPO_ROOT/tools/run_c4.py 30 > /dev/null 2>&1; f=$(find /tmp/p_c4 -name "*kernel_stats.csv" | head -1); grep "cma_gemm\|cov_finish\|symmetrize\|cma_y" $f | cut -c1-200
