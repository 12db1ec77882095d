#!/bin/bash
# GPU box: one rank, the same workload through the single-GPU path, the peer exchange and the rccl transport
cd "${GRAFT_REPO_ROOT:-.}"
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("%-30s %-8s %8.2f us/step" % (d["config"]["workload"], sys.argv[1], d["ms_per_step"]*1e3))'
for wl in ${WL:-de_rosenbrock_n1024_p16384}; do
  timeout 300 python bench.py --no-cpu-baseline --workload $wl --steps ${STEPS:-400} --warmup 100 2>&1 < /dev/null | tail -1 | python -c "$fmt" single
  for ex in p2p rccl; do
    SX_FORCE_SHARDED=1 SX_EXCHANGE=$ex timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
      --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline --workload $wl \
      --steps ${STEPS:-400} --warmup 100 2>&1 < /dev/null | tail -1 | python -c "$fmt" $ex
  done
done
