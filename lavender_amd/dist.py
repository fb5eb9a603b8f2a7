"""Process-group helpers -- mirror of the reference's utils/dist.py:20-129 for the hot path (init from the
launcher environment, seeds, rank getters, scalar reduce).  Backend "nccl" is RCCL on ROCm; "gloo" is accepted so
the data-parallel logic can be exercised on CPU boxes."""
import os
import random
from datetime import timedelta

import numpy as np
import torch
import torch.distributed as dist


def dist_init(args, distributed=True, backend=None):
    """utils/dist.py:20-75: OMPI_* or torchrun env -> init_process_group; no launcher -> single process."""
    if distributed and 'OMPI_COMM_WORLD_SIZE' in os.environ:
        world_size = int(os.environ['OMPI_COMM_WORLD_SIZE'])
        rank = int(os.environ['OMPI_COMM_WORLD_RANK'])
        args.local_rank = int(os.environ['OMPI_COMM_WORLD_LOCAL_RANK'])
        args.num_gpus = world_size
        args.distributed = world_size > 1
        args.num_nodes = world_size // 8
        if args.distributed:
            addr, port = os.environ.get("MASTER_ADDR", '127.0.0.1'), os.environ.get("MASTER_PORT", 12345)
            _set_device(args.local_rank)
            dist.init_process_group(backend=backend or _default_backend(), init_method=f"tcp://{addr}:{port}",
                                    world_size=world_size, rank=rank, timeout=timedelta(hours=5))
    elif distributed and 'WORLD_SIZE' in os.environ:
        args.num_gpus = int(os.environ['WORLD_SIZE'])
        args.local_rank = int(os.environ.get('LOCAL_RANK', 0))
        args.distributed = True
        _set_device(args.local_rank)
        if not dist.is_initialized():
            dist.init_process_group(backend=backend or _default_backend(), init_method='env://', timeout=timedelta(hours=5))
    else:
        print("no distributed training ... presumbly debug with 1 GPU")
        args.num_gpus = 1
        args.distributed = False
    set_seed(getattr(args, "seed", 88), args.num_gpus)


def _default_backend():
    return "nccl" if torch.cuda.is_available() else "gloo"


def _set_device(local_rank):
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)


def set_seed(seed, n_gpu=1):
    """utils/dist.py:78-83 (same seed on every rank)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    from . import hip
    hip.reseed(seed)


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def get_local_rank():
    return int(os.environ.get("LOCAL_RANK", os.environ.get("OMPI_COMM_WORLD_LOCAL_RANK", 0)))


def is_main_process():
    return get_rank() == 0


def synchronize():
    if get_world_size() > 1:
        dist.barrier()
