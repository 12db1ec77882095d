#!/bin/bash
# Round 6: the resident form of the eigensolver against the launch per round, BASELINE config 4 and the solver alone.
# usage (on the GPU box): tools/r6_eigh_flow.sh > gpurun_out/r6_eigh_flow_runs.txt
export SX_EIGH_FLOW_TIMEOUT_MS=500
for g in 0 1 2 3; do for w in 128 240; do
  echo "== SX_EIGH_FLOW_GRAN=$g (bit 0: rotations as tagged halves instead of slot words; bit 1: the next round's loads behind the drain) SX_EIGH_FLOW_WORKERS=$w"
  SX_EIGH_FLOW_GRAN=$g SX_EIGH_FLOW_WORKERS=$w timeout 300 python tools/bench_eigh_flow.py 512 1024 3 2>&1 | grep -v -i "warn\|amdgpu.ids" | tail -6
done; done
echo "== timeline of the resident kernel (default switches, 128 workers): tools/trace_eigh_flow.py 512 40"
timeout 300 python tools/trace_eigh_flow.py 512 40 2>&1 | grep -v amdgpu.ids
