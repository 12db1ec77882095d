#!/bin/bash
# round 4, second half: long rows with numpy's plan as constants -- whole GPU suite, sx_eval, the bench's configurations
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4b
timeout 2400 python -m pytest tests -m gpu -q -x -n 4 2>&1 | tail -3
timeout 300 python tools/bench_eval.py 2>&1 | grep sx_eval > gpurun_out/r4b/eval_long_rows.txt
cat gpurun_out/r4b/eval_long_rows.txt
timeout 600 python bench.py --steps 2000 --warmup 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('M', d['ms_per_step'], d['roofline']['frac'])
for k,v in d.get('configs',{}).items(): print(k, {a:b for a,b in v.items() if a!='note'})
" > gpurun_out/r4b/bench_configs.txt 2>&1
cat gpurun_out/r4b/bench_configs.txt
