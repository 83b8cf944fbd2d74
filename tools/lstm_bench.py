"""Recurrent step per level at config-A shapes: fused split-bf16 step (lstm_sb.h) vs the fp32 matrix-core step (lstm16.h):
HIP-event time per step launch and the difference of the hidden sequences.   GPU box:  python tools/lstm_bench.py [H W]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bde2vid_amd import canonical, ops, _lib
from bde2vid_amd.model import build_model
from bde2vid_amd.weights import formula_state_dict
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (184, 240)
T = 16
cfg = canonical()
m = build_model(cfg, formula_state_dict(cfg), 'cuda:0')
L = _lib.lib()
g = torch.Generator(device='cuda').manual_seed(0)
for l in range(3):
    x = torch.randn(T, 1, cfg.enc_in(l), H >> l, W >> l, device='cuda', generator=g)
    res = {}
    for mode in (0, 1, 0, 1):
        m.set_tuning('lstm_sbk', mode)
        h, c = ops.recurrent_conv(m, l, 0, x)
        L.bde_profile_reset(m._h, 1)
        for _ in range(3):
            ops.recurrent_conv(m, l, 0, x)
        torch.cuda.synchronize()
        ms, cnt = C.c_double(), C.c_int64()
        L.bde_profile_get(m._h, f'lstm{l}'.encode(), C.byref(ms), C.byref(cnt))
        L.bde_profile_reset(m._h, 0)
        res[mode] = (h, c, ms.value / cnt.value * 1e3)
    d = float((res[1][0] - res[0][0]).abs().max())
    dc = float((res[1][1] - res[0][1]).abs().max())
    print(f'level {l}: lstm16 {res[0][2]:.1f} us / step, lstm_sb {res[1][2]:.1f} us / step; max |dh| {d:.2e}, max |dc| {dc:.2e}, max |h| {float(res[0][0].abs().max()):.2f}')

import numpy as np
L.bde_debug_token_stamps(m._h, None, 0)
m.set_tuning('lstm_sbk', 1)
for l in range(3):
    x = torch.randn(T, 1, cfg.enc_in(l), H >> l, W >> l, device='cuda', generator=g)
    ops.recurrent_conv(m, l, 0, x)
    torch.cuda.synchronize()
    out = (C.c_int64 * 2048)()
    L.bde_debug_token_stamps(m._h, out, 2048)
    print(f'level {l}: first workgroup start -> last workgroup end: {(out[2047] - out[2046]) / 100.0:.1f} us (all workgroups, s_memrealtime)')
    a = np.array(out[:], dtype=np.int64).reshape(64, 4, 8)
    d = a[:, :, 1:7] - a[:, :, 0:6]
    print(f'level {l} phase cycles (prologue, first DMA, stages, barrier, reduce, tail+store) median:', np.median(d.reshape(-1, 6), axis=0),
          ' total', np.median(a[:, :, 6] - a[:, :, 0]), ' in-kernel clock GHz', round(float(np.median((a[:, :, 6] - a[:, :, 0]) / np.maximum(a[:, :, 7], 1))) * 0.1, 3), ' spread of start', int(a[:, :, 0].max() - a[:, :, 0].min()), ' end', int(a[:, :, 6].max() - a[:, :, 6].min()))

m.set_tuning('tok_debug', 9)
for l in range(3):
    x = torch.randn(T, 1, cfg.enc_in(l), H >> l, W >> l, device='cuda', generator=g)
    ops.recurrent_conv(m, l, 0, x)
    torch.cuda.synchronize()
    out = (C.c_int64 * 2048)()
    L.bde_debug_token_stamps(m._h, out, 2048)
    nwg = {0: 692, 1: 704, 2: 512}[l] if (H, W) == (184, 240) else 1000
    a = np.array(out[:2000], dtype=np.int64).reshape(1000, 2)[:nwg]
    a = a[a[:, 1] > 0]
    t0 = a[:, 0].min()
    st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0
    print(f'level {l}: {len(a)} workgroups; start us: min {st.min():.1f} median {np.median(st):.1f} p90 {np.percentile(st, 90):.1f} max {st.max():.1f}; '
          f'duration us: median {np.median(en - st):.1f} max {(en - st).max():.1f}; end max {en.max():.1f}; started after 5 us: {(st > 5).sum()}')
m.set_tuning('tok_debug', 0)
