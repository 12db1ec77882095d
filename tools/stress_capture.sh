#!/bin/bash
# the RCCL graph-capture tests N times in fresh processes (each captures collectives of a live
# nccl process group while torch's watchdog thread runs); counts aborts.  usage: stress_capture.sh [N]
N=${1:-200}
ok=0; bad=0
for i in $(seq 1 $N); do
  if timeout 300 python -m pytest tests/test_distributed.py -q -x -k "rccl_graph_capture" -p no:cacheprovider > /tmp/cap_$i.log 2>&1; then ok=$((ok+1)); else bad=$((bad+1)); tail -5 /tmp/cap_$i.log; fi
done
echo "capture stress: $ok ok, $bad failed of $N runs"
