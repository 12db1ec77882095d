#!/bin/bash
# Round 5, after the three-launch best / termination step for wide rows: GPU suite, sharded fuzzer, the wide-row table, kernel stats
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5l; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/pytest_gpu.txt
python tools/fuzz_sharded.py 2>&1 | tail -2 > $O/fuzz_sharded.txt
timeout 900 python tools/bench_wide.py 2>&1 | grep -v amdgpu.ids > $O/wide_rows.txt
PROF_LINES=14 bash tools/prof_cmd.sh widede2 $PWD/tools/bench_wide.py de pso > /dev/null 2>&1
cp $(find gpurun_out/prof_widede2 -name "*kernel_stats.csv" | head -1) $O/wide_de_pso_kernel_stats.csv
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > $O/smoke.txt
cat $O/pytest_gpu.txt $O/fuzz_sharded.txt $O/smoke.txt $O/wide_rows.txt; head -8 $O/wide_de_pso_kernel_stats.csv | cut -c1-160
