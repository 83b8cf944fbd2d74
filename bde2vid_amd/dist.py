"""Multi-GPU layer: one process per GPU, sequences sharded across ranks, ONE collective.

The reference has no distributed code.  Sequences (<= subseq_L chunks,
eval_models_seq.py:216-219) are independent because BDE2VID.forward resets its state at every
call (bde2vid.py:31), so the only exchange is a single RCCL broadcast of the packed weight image
(about 100 MB at config A) over xGMI at start-up; nothing is communicated per step.
"""
import os
from typing import List, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: str = None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun); no-op single process."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1 and not os.environ.get('BDE_FORCE_DIST'):   # (BDE_FORCE_DIST: exercise RCCL with one rank)
        return 0, 1, 0
    os.environ.setdefault('RANK', '0')
    os.environ.setdefault('MASTER_PORT', '29500')
    rank = int(os.environ['RANK'])
    local = int(os.environ.get('LOCAL_RANK', rank))
    if backend is None:
        backend = os.environ.get('BDE_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if torch.cuda.is_available():
        local = local % max(torch.cuda.device_count(), 1)   # (functional tests put two ranks on one GPU over gloo)
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_sequences(n_sequences: int, rank: int, world: int) -> List[int]:
    """Round-robin assignment of independent sequences to ranks."""
    return list(range(rank, n_sequences, world))


def broadcast_packed(flat: torch.Tensor, src: int = 0) -> torch.Tensor:
    """Broadcast a flat fp32 weight image in place (RCCL on GPUs, gloo in the CPU tests)."""
    if dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get('BDE_FORCE_DIST')):
        dist.broadcast(flat, src=src)
    return flat


def build_replicated_model(cfg, state_dict_fn, device):
    """Rank 0 packs the weights (state_dict_fn() is only called there); every other rank
    allocates the same packed layout and receives the image by one broadcast."""
    from .model import BDE2VID
    rank = dist.get_rank() if dist.is_initialized() else 0
    m = BDE2VID(generator=cfg).to(device)
    if rank == 0:
        m.load_state_dict(state_dict_fn())
    else:
        m.alloc_packed()
    if dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get('BDE_FORCE_DIST')):
        broadcast_packed(m.packed_view(), 0)
        torch.cuda.synchronize(device)
    return m


def max_over_ranks(value: float, device=None) -> float:
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if dist.get_backend() == 'nccl' else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def min_over_ranks(value: float, device=None) -> float:
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if dist.get_backend() == 'nccl' else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
