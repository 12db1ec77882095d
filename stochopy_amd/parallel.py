"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend nccl = RCCL)."""


def require_world(workers):
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        raise RuntimeError(
            f"workers={workers}: launch one process per GPU with torchrun and call "
            "torch.distributed.init_process_group first (see bench.py)")
    if dist.get_world_size() != workers:
        raise RuntimeError(f"workers={workers} but the process group has {dist.get_world_size()} ranks")
