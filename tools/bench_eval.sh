#!/bin/bash
cd "$(dirname "$0")/.."
python tools/bench_eval.py 2>&1 | grep -v amdgpu.ids
