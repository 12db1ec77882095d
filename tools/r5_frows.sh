#!/bin/bash
# round 5: SURVEY 8(f) rows on the current binary -- whole-generation lines, then kernel stats of each family under rocprofv3
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5
python tools/time_vd_minimize.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/vd_minimize_times.txt
PROF_LINES=24 LOG_LINES=2 bash tools/prof_cmd.sh r5_vd_wide $PWD/tools/run_vd_wide.py 16384 1024 40 2>&1 | tee gpurun_out/r5/vd_wide_p1024_stats.txt
PROF_LINES=24 LOG_LINES=2 bash tools/prof_cmd.sh r5_vd_wide4k $PWD/tools/run_vd_wide.py 16384 4096 20 2>&1 | tee gpurun_out/r5/vd_wide_p4096_stats.txt
PROF_LINES=14 LOG_LINES=4 bash tools/prof_cmd.sh r5_immediate $PWD/tools/r5_frows.py immediate 2>&1 | tee gpurun_out/r5/immediate_stats.txt
PROF_LINES=14 LOG_LINES=4 bash tools/prof_cmd.sh r5_na $PWD/tools/r5_frows.py na 2>&1 | tee gpurun_out/r5/na_stats.txt
