"""winblock_kernel at level 0 of config A (92x120 map): HIP-event time per launch, phase stamps, parity vs the split path.
GPU box:  python tools/win_bench.py [H W]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from bde2vid_amd import canonical, ops, _lib
from bde2vid_amd.model import build_model
from bde2vid_amd.weights import formula_state_dict
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (92, 120)
cfg = canonical()
m = build_model(cfg, formula_state_dict(cfg), 'cuda:0')
L = _lib.lib()
g = torch.Generator(device='cuda').manual_seed(1)
bufs = [torch.randn(1, 64, H, W, device='cuda', generator=g) for _ in range(3)]
for _ in range(3):
    y = ops.dframe_attention(m, 0, bufs)
torch.cuda.synchronize()
m.set_tuning('winblock', 0)
y_split = ops.dframe_attention(m, 0, bufs)
m.set_tuning('winblock', 1)
print('max |winblock - split path| over 4 blocks:', float((y - y_split).abs().max()))
for sb in (0, 1, 0, 1):
    m.set_tuning('winblock_sb', sb)
    ysb = ops.dframe_attention(m, 0, bufs)
    L.bde_profile_reset(m._h, 1)
    for _ in range(25):
        ops.dframe_attention(m, 0, bufs)
    torch.cuda.synchronize()
    ms, cnt = C.c_double(), C.c_int64()
    L.bde_profile_get(m._h, b'winblock0', C.byref(ms), C.byref(cnt))
    print(f'winblock_sb={sb}: {cnt.value} launches, {ms.value / cnt.value * 1e3:.2f} us each (HIP events, eager); '
          f'max |y - split path| {float((ysb - y_split).abs().max()):.3e}')
    L.bde_profile_reset(m._h, 0)
L.bde_debug_token_stamps(m._h, None, 0)
for sb in (0, 1):
    m.set_tuning('winblock_sb', sb)
    ops.dframe_attention(m, 0, bufs, 0, 1)
    torch.cuda.synchronize()
    out = (C.c_int64 * 2048)()
    L.bde_debug_token_stamps(m._h, out, 2048)
    a = np.array(out[:], dtype=np.int64).reshape(64, 4, 8)
    d = a[:, :, 1:7] - a[:, :, 0:6]
    print(f'winblock_sb={sb} phase cycles (gather, qkv, attention, proj, fc1, fc2), median over (block, wave):',
          np.median(d.reshape(-1, 6), axis=0), ' total', np.median(a[:, :, 6] - a[:, :, 0]),
          ' score loop (stamp 7 - stamp 2)', np.median(a[:, :, 7] - a[:, :, 2]))
