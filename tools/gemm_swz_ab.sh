#!/bin/bash
# round 4: the sampling GEMM's workgroup numbering (row tile fastest: an XCD meets an eighth of Z and all of B D) against the
# launch order (column tile fastest) -- isolated launches and in situ (rocprofv3 over a C4 run)
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/gemm_swz_ab.txt; : > $out
for rep in 1 2; do
  for lib in stochopy_amd/lib/libstochopy_hip.so build_ab/libsx_gemm_noswz.so; do
    echo "== $lib" >> $out
    python tools/ab_lib.py $lib tools/bench_cma_kernels.py 512 1024 1024 2048 256 512 2>&1 | grep -i "sample" | grep -v vdsample >> $out
  done
done
for lib in stochopy_amd/lib/libstochopy_hip.so build_ab/libsx_gemm_noswz.so; do
  echo "== in situ, $lib" >> $out
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p_swz && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_swz -o run -- python $GRAFT_REPO_ROOT/tools/ab_lib.py $GRAFT_REPO_ROOT/$lib $GRAFT_REPO_ROOT/tools/run_c4.py 40 > /dev/null 2>&1; f=$(find /tmp/p_swz -name "*kernel_stats.csv" | head -1); grep "cma_gemm_kernel<0" $f | awk -F'",' '{print "sampling GEMM in situ (calls,total ns,avg ns,...):", $NF}') >> $out
done
timeout 600 python -m pytest tests/test_gpu_cmaes.py -q -x -k "sample or golden or oracle" 2>&1 | tail -2 >> $out
cat $out
