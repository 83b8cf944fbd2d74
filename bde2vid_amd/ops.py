"""Single sub-modules of the reference, bound to the C ABI's bde_op_* entry points.

Used by the per-block parity tests (they read like the reference's sub-module calls in
oracle/gen_golden.py) and by nothing on the whole-model path.
"""
import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib
from .model import BDE2VID, _stream_ptr


def _chk(t: torch.Tensor):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise TypeError('expected a contiguous float32 CUDA tensor')
    return t


def _sp(m):
    return C.c_void_p(_stream_ptr(m.device))


def head(m: BDE2VID, x):
    _chk(x)
    N, _, H, W = x.shape
    out = torch.empty((N, m.cfg.basechannels, H, W), device=x.device)
    _lib.check(_lib.lib().bde_op_head(m._h, C.c_void_p(x.data_ptr()), N, H, W, C.c_void_p(out.data_ptr()), _sp(m)))
    return out


def encoder_conv(m: BDE2VID, level, direction, x):
    _chk(x)
    N, _, H, W = x.shape
    out = torch.empty((N, m.cfg.enc_out(level), H // 2, W // 2), device=x.device)
    _lib.check(_lib.lib().bde_op_encoder_conv(m._h, level, direction, C.c_void_p(x.data_ptr()), N, H, W,
                                              C.c_void_p(out.data_ptr()), _sp(m)))
    return out


def recurrent_conv(m: BDE2VID, level, direction, x):
    """x: [T,B,Cin,H,W] -> (h [T,B,C,H/2,W/2], c [B,C,H/2,W/2])."""
    _chk(x)
    T, B, _, H, W = x.shape
    Cc = m.cfg.enc_out(level)
    h = torch.empty((T, B, Cc, H // 2, W // 2), device=x.device)
    c = torch.empty((B, Cc, H // 2, W // 2), device=x.device)
    _lib.check(_lib.lib().bde_op_recurrent_conv(m._h, level, direction, C.c_void_p(x.data_ptr()), T, B, H, W,
                                                C.c_void_p(h.data_ptr()), C.c_void_p(c.data_ptr()), _sp(m)))
    return h, c


def gate_conv(m: BDE2VID, level, x):
    """x: [2, N, C, H, W] (forward / backward encoder outputs) -> [2, N, 4C, H, W]: W_x * x + bias of the gates."""
    _chk(x)
    _, N, Cc, H, W = x.shape
    out = torch.empty((2, N, 4 * Cc, H, W), device=x.device)
    _lib.check(_lib.lib().bde_op_gate_conv(m._h, level, C.c_void_p(x.data_ptr()), N, H, W, C.c_void_p(out.data_ptr()), _sp(m)))
    return out


def decoder(m: BDE2VID, j, x, skip: Optional[torch.Tensor] = None):
    _chk(x)
    N, _, H, W = x.shape
    ne = m.cfg.num_encoders
    out = torch.empty((N, m.cfg.enc_in(ne - 1 - j), 2 * H, 2 * W), device=x.device)
    sp = C.c_void_p(_chk(skip).data_ptr()) if skip is not None else None
    _lib.check(_lib.lib().bde_op_decoder(m._h, j, C.c_void_p(x.data_ptr()), sp, N, H, W,
                                         C.c_void_p(out.data_ptr()), _sp(m)))
    return out


def pred(m: BDE2VID, x, head_feat: Optional[torch.Tensor] = None):
    _chk(x)
    N, _, H, W = x.shape
    out = torch.empty((N, 1, H, W), device=x.device)
    hp = C.c_void_p(_chk(head_feat).data_ptr()) if head_feat is not None else None
    _lib.check(_lib.lib().bde_op_pred(m._h, C.c_void_p(x.data_ptr()), hp, N, H, W, C.c_void_p(out.data_ptr()), _sp(m)))
    return out


def dframe_attention(m: BDE2VID, level, bufs: Sequence[Optional[torch.Tensor]], first_block=0, nblocks=-1):
    """bufs: frame_num tensors [B,C,H,W]; None = all-zero frame."""
    q = _chk(bufs[m.cfg.q_idx])
    B, Cc, H, W = q.shape
    ptrs = (C.c_void_p * len(bufs))(*[(_chk(b).data_ptr() if b is not None else None) for b in bufs])
    out = torch.empty_like(q)
    _lib.check(_lib.lib().bde_op_dframe_attention(m._h, level, ptrs, B, H, W, first_block, nblocks,
                                                  C.c_void_p(out.data_ptr()), _sp(m)))
    return out
