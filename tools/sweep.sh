#!/bin/bash
# GPU box: DE shape sweep (bench.py workloads) + the other configs; one JSON line / text line each.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
: > gpurun_out/sweep.log
for wl in ${SWEEP:-de_rosenbrock_n128_p4096 de_rosenbrock_n256_p4096 de_rosenbrock_n512_p8192 de_rosenbrock_n1024_p16384 de_rastrigin_n1024_p16384 de_rosenbrock_n2048_p16384}; do
  timeout 300 python bench.py --workload $wl --steps 400 --warmup 100 --no-cpu-baseline 2>&1 < /dev/null | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-32s %8.2f us/step  kernel %8.2f us  %7.1f GB/s  frac %.3f' % (d['config']['workload'], d['ms_per_step']*1e3, r['kernel_us'], r['achieved'], r['frac']))" >> gpurun_out/sweep.log 2>&1
done
if [ -z "$NO_OTHER" ]; then timeout 600 python tools/bench_other.py >> gpurun_out/sweep.log 2>&1 < /dev/null; fi
cat gpurun_out/sweep.log
