#!/usr/bin/env python3
"""Isolated timing of single stages at config-A shapes (used under rocprofv3 --pmc as well).
usage: tools/microbench.py <stage> [reps]   stage in: lstm0 lstm1 lstm2 attn0 attn2 dec head all"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bde2vid_amd import canonical, ops
from bde2vid_amd.model import build_model
from bde2vid_amd.weights import formula_state_dict

stage = sys.argv[1] if len(sys.argv) > 1 else 'lstm0'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = canonical()
m = build_model(cfg, formula_state_dict(cfg), 'cuda:0')
if os.environ.get('PW_FORCE'):
    m.set_tuning('pw_force', int(os.environ['PW_FORCE']))
if os.environ.get('CONV_VEC'):
    m.set_tuning('conv_vec', int(os.environ['CONV_VEC']))
if os.environ.get('LSTM_SHAPE'):
    m.set_tuning('lstm_shape', int(os.environ['LSTM_SHAPE']))
if os.environ.get('CONV_NT'):
    m.set_tuning('conv_nt', int(os.environ['CONV_NT']))
if os.environ.get('LSTM_HC8'):
    m.set_tuning('lstm_hc8', int(os.environ['LSTM_HC8']))
if os.environ.get('TOK_NPT'):
    m.set_tuning('tok_npt', int(os.environ['TOK_NPT']))
if os.environ.get('TOK_DEBUG'):
    m.set_tuning('tok_debug', int(os.environ['TOK_DEBUG']))
if os.environ.get('FUSED_MIN'):
    m.set_tuning('fused_min_tiles', int(os.environ['FUSED_MIN']))
H, W, T = 184, 240, 16
g = torch.Generator(device='cuda').manual_seed(0)


def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


if stage.startswith('lstm'):
    l = int(stage[4:])
    x = torch.randn(T, 1, cfg.enc_in(l), H >> l, W >> l, device='cuda', generator=g)
    print(stage, 'ms per 16-step sweep (both dirs, incl. enc+gx convs):', timed(lambda: ops.recurrent_conv(m, l, 0, x)))
elif stage.startswith('attn'):
    l = int(stage[4:])
    C = cfg.enc_out(l)
    bufs = [torch.randn(1, C, H >> (l + 1), W >> (l + 1), device='cuda', generator=g) for _ in range(3)]
    print(stage, 'ms per frame (all blocks):', timed(lambda: ops.dframe_attention(m, l, bufs)))
elif stage == 'dec':
    for j in range(3):
        l = 2 - j
        x = torch.randn(T, cfg.enc_out(l), H >> (l + 1), W >> (l + 1), device='cuda', generator=g)
        print('dec', j, 'ms per 16 frames:', timed(lambda: ops.decoder(m, j, x, x)))
elif stage == 'head':
    x = torch.randn(T, 5, H, W, device='cuda', generator=g)
    print('head ms per 16 frames:', timed(lambda: ops.head(m, x)))
