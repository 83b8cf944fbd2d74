"""Multi-GPU layer: one process per GPU, sequences sharded across ranks, ONE collective.

The reference has no distributed code.  Sequences (<= subseq_L chunks,
eval_models_seq.py:216-219) are independent because BDE2VID.forward resets its state at every
call (bde2vid.py:31), so the only exchange is a single RCCL broadcast of the packed weight image
(about 100 MB at config A) over xGMI at start-up; nothing is communicated per step.
"""
import ctypes as C
import os
from typing import List, Optional, Sequence

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


class DirectionSplit:
    """ONE sequence on TWO GPUs (SURVEY.md §8e option 1): the forward and the backward RecurrentConv sweeps of a level are
    independent until their outputs are added (V5.py:122-147), so rank `ranks[0]` runs the forward sweeps, the merge, the temporal
    attention and the decoder, rank `ranks[1]` the backward sweeps.  Per level two tensors cross the link: the backward hidden
    sequence before the merge and the level's refined output as the next level's input ([T, B, C, h, w] fp32 each; at 720x1280,
    T = 64 that is 3.8 + 3.8 GB for level 0 -- xGMI carries one such transfer in ~30 ms, so the split pays only while a
    level's sweep takes longer than its two transfers).  Both ranks call `forward` with the same inputs; the first rank returns
    the frames (bit-identical to `model(inputs)` on one GPU: every launch is the joint forward's launch for that direction),
    the second returns None.

    `group` is a process group holding exactly the two ranks (None = the default group, which must then have two ranks).
    The exchange is a two-rank broadcast: one point-to-point copy over xGMI under RCCL, and the one collective gloo also
    offers for device tensors (the functional test puts both ranks on one GPU)."""

    def __init__(self, model, ranks: Sequence[int] = (0, 1), group=None):
        if len(ranks) != 2 or ranks[0] == ranks[1]:
            raise ValueError('DirectionSplit takes two different ranks')
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self.model, self.ranks, self.group = model, tuple(int(r) for r in ranks), group
        me = dist.get_rank()
        if me not in self.ranks:
            raise RuntimeError(f'rank {me} is not one of {self.ranks}')
        self.direction = self.ranks.index(me)                 # 0: forward sweeps + everything else, 1: backward sweeps

    def _buffer(self, what: str, level: int, direction: int) -> torch.Tensor:
        from . import _lib
        ptr, n = C.c_void_p(), C.c_int64()
        _lib.check(_lib.lib().bde_split_buffer(self.model._h, what.encode(), level, direction, C.byref(ptr), C.byref(n)))

        class _Holder:
            pass
        h = _Holder()
        h.__cuda_array_interface__ = {'shape': (int(n.value),), 'typestr': '<f4', 'data': (int(ptr.value), False), 'version': 3}
        t = torch.as_tensor(h, device=self.model.device)
        t._bde_keepalive = h
        return t

    def _move(self, t: torch.Tensor, src_index: int):
        dist.broadcast(t, src=self.ranks[src_index], group=self.group)

    def forward(self, inputs: Sequence[dict]) -> Optional[List[torch.Tensor]]:
        from . import _lib
        from .model import _stream_ptr
        m, L = self.model, _lib.lib()
        evs = [d['events'].contiguous() for d in inputs]
        T = len(evs)
        B, nb, H, W = evs[0].shape
        dev = evs[0].device
        if nb != m.cfg.num_bins or any(tuple(e.shape) != (B, nb, H, W) or e.dtype != torch.float32 or not e.is_cuda for e in evs):
            raise ValueError('all frames must be float32 CUDA tensors of shape [B, num_bins, Hp, Wp]')
        if m.get_info('pipeline') != 1:
            raise RuntimeError('DirectionSplit runs on the caller\'s stream: set_tuning("pipeline", 1)')
        ne, d = m.cfg.num_encoders, self.direction
        with torch.cuda.device(dev):
            st = C.c_void_p(_stream_ptr(dev))
            ev_ptrs = (C.c_void_p * T)(*[e.data_ptr() for e in evs])
            _lib.check(L.bde_split_begin(m._h, ev_ptrs, T, B, H, W, st))
            for l in range(ne):
                _lib.check(L.bde_split_sweep(m._h, l, d, st))
                self._move(self._buffer('hidden', l, 1), 1)              # backward hidden sequence -> rank A
                if d == 0:
                    _lib.check(L.bde_split_attend(m._h, l, st))
                if l + 1 < ne:
                    self._move(self._buffer('level_out', l, 0), 0)       # refined level output -> rank B
            if d != 0:
                return None
            out = torch.empty((T, B, 1, H, W), dtype=torch.float32, device=dev)
            im_ptrs = (C.c_void_p * T)(*[out[t].data_ptr() for t in range(T)])
            _lib.check(L.bde_split_decode(m._h, im_ptrs, st))
        return [out[t] for t in range(T)]

    __call__ = forward
