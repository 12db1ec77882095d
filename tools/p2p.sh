#!/bin/bash
# GPU box: multi-process tests of both exchanges + the bench through each exchange with one rank.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_distributed.py -x -q -m gpu > gpurun_out/p2p_tests.log 2>&1 < /dev/null
echo "tests rc=$?" >> gpurun_out/p2p_tests.log
tail -25 gpurun_out/p2p_tests.log
for ex in p2p rccl; do
  SX_FORCE_SHARDED=1 SX_EXCHANGE=$ex timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
    --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline \
    > gpurun_out/bench_$ex.log 2>&1 < /dev/null
  echo "bench $ex rc=$?"
  tail -2 gpurun_out/bench_$ex.log
done
