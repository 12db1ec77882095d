#!/bin/bash
# A/B of the eigensolver's deep refinement step inside BASELINE config 4 (SX_EIGH_DEEP=0: the round-5 form, a whole sweep where
# the deep step now stands), alternating on one box; kernel statistics of both.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do for d in 1 0; do echo "SX_EIGH_DEEP=$d: $(SX_EIGH_DEEP=$d python $R/tools/bench_c4.py 10 60 2>&1 | grep 'device-resident loop:' | head -1)"; done; done
for d in 1 0; do
  SX_EIGH_DEEP=$d rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/deep$d -o run -- python $R/tools/run_c4.py 60 > /dev/null 2>&1
  echo "== kernel statistics, SX_EIGH_DEEP=$d (60 generations): name, calls, average ns, % of GPU time"
  f=$(find /tmp/deep$d -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
for r in rows[1:9]:
    print("   %-70s %6s %10.0f %6s" % (r[0][:70], r[1], float(r[3]), r[4][:5]))
PY
done
