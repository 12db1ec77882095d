#!/bin/bash
# generic A/B at the metric shape: tools/m_ab.sh lib1.so lib2.so ... (three alternating repetitions)
cd "$(dirname "$0")/.."
for rep in 1 2 3; do for lib in "$@"; do python tools/ab_lib.py $lib tools/nt_ab.py de 2>&1 | grep -v amdgpu.ids; done; done
