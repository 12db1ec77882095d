cd $GRAFT_REPO_ROOT
SX_BENCH_DEVICE=0 SX_BENCH_BACKEND=gloo timeout 800 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --kernel-timing-launches 50 2>&1 | tail -3 | cut -c1-3000
