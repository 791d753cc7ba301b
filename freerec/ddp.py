"""`freerec.ddp`: the distributed helpers the Coach uses (one process per GPU under torchrun; RCCL is torch's "nccl" backend)."""
import functools
import os

import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_local_rank():
    return int(os.environ.get("LOCAL_RANK", 0))


def is_main_process():
    return get_rank() == 0


def main_process_only(fn):
    @functools.wraps(fn)
    def wrapper(*a, **k):
        return fn(*a, **k) if is_main_process() else None
    return wrapper


def synchronize():
    if is_distributed():
        dist.barrier()
